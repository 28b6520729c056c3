"""The N>1 path on CPU: two gloo ranks shard a clip list, agree on timing and cover every clip exactly once."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mudg_amd import parallel
    r, w, _, dist = parallel.init_from_env("gloo")
    mine = parallel.shard_clips(11, r, w)
    parallel.barrier(dist)
    slow = parallel.max_over_ranks(1.0 + r, dist)
    total = parallel.sum_over_ranks(len(mine), dist)
    seeds = [parallel.clip_seed(123, c) for c in mine]
    out.put((r, mine, slow, total, seeds))
    dist.destroy_process_group()


def test_two_ranks_shard_clips_without_overlap():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    clips = sorted(c for _, mine, _, _, _ in got for c in mine)
    assert clips == list(range(11))
    assert all(slow == 2.0 and total == 11.0 for _, _, slow, total, _ in got)
    seeds = {c: s for _, mine, _, _, ss in got for c, s in zip(mine, ss)}
    from mudg_amd import parallel
    assert seeds == {c: parallel.clip_seed(123, c) for c in range(11)}       # independent of the sharding


def test_single_process_is_a_no_op():
    from mudg_amd import parallel
    assert parallel.shard_clips(5, 0, 1) == [0, 1, 2, 3, 4]
    assert parallel.max_over_ranks(3.5, None) == 3.5


def _bench_worker(rank, world, port, q):
    """bench.main() as the driver launches it for N > 1 (torchrun environment, one process per GPU), with the device set-up and
    the workload swapped for CPU stubs: what is under test is the rank logic — per-rank clip seed, rank-0-only build behind a
    barrier, barrier + max-over-ranks timing, ONE JSON line from rank 0 and silence from the others."""
    import contextlib, io, json, time
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--no-decode", "--no-cpu-baseline", "--no-children"]
    import torch
    import bench
    from mudg_amd import parallel
    seen = {}

    def make_workload(args, device, r, dist):
        seen["built_by"] = r if r == 0 else None              # the real one builds on rank 0 only, then everybody meets at a barrier
        parallel.barrier(dist)
        seen["seed"] = parallel.clip_seed(123, r)

        def run(n, x, start_index):
            time.sleep(0.05 * n * (1 + r))                      # rank 1 is twice as slow: the reported time must be ITS time
            return x + n
        return {"run": run, "x0": torch.zeros(2, 3), "S": 50, "model": None, "inp": None, "hip": None}

    bench.setup = lambda args, local: (torch.device("cpu"), "gloo")
    bench.make_workload = make_workload
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    rec = json.loads(lines[0]) if lines else None
    q.put((rank, len(lines), rec, seen["seed"]))


def test_bench_rank_logic_over_two_gloo_ranks():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    (r0, n0, rec, seed0), (r1, n1, none, seed1) = res
    assert (n0, n1) == (1, 0) and none is None                       # ONE JSON line, from rank 0
    assert seed0 != seed1                                            # every rank denoises its own clip
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    # 3 steps at 0.1 s on the slow rank: the max over ranks, and value = N x steps / that time (whole-job aggregate)
    total_ms = rec["ms_per_step"] * rec["steps"]
    assert 290 < total_ms < 600                                      # the slow rank's 0.3 s, not the fast rank's 0.15 s
    assert abs(rec["value"] - 2 * rec["steps"] / (total_ms / 1000)) < 0.05 * rec["value"]
