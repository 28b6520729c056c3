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
