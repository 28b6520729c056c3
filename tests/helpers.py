"""Shared test helpers: golden fixtures, seeded weights, error metrics."""
import importlib.util
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    spec = importlib.util.spec_from_file_location("golden_" + name, os.path.join(GOLDEN, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


seeding = _load("seeding")
cfgs = _load("configs")


def golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=True)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def seeded_sd(param_shapes, seed, expect_checksum=None):
    sd = seeding.seeded_state_dict({k: tuple(v) for k, v in param_shapes.items()}, seed)
    if expect_checksum is not None:
        got = seeding.checksum(sd)
        assert abs(got - expect_checksum) <= 1e-9 * abs(expect_checksum), (got, expect_checksum)
    return sd


def unet_inputs(cfg, shp, seed):
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    x = seeding.seeded_input("x", (B, cfg["in_channels"], T, H, W), seed)
    ctx = seeding.seeded_input("context", (B, 77 + 16 * T, cfg["context_dim"]), seed)
    return x, ctx


def pipeline_inputs(g, steps=None):
    """Rebuild the tensors make_golden.golden_pipeline fed the reference."""
    shp, s, seed = g["shape"], dict(g["sampler"]), g["seed"] + 2
    if steps is not None:
        s["steps"] = steps
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    cd = g["unet_cfg"]["context_dim"]
    return {
        "ctx_c": seeding.seeded_input("ctx_cond", (B, 77 + 16 * T, cd), seed),
        "ctx_u": seeding.seeded_input("ctx_uncond", (B, 77 + 16 * T, cd), seed),
        "concat": seeding.seeded_input("c_concat", (B, 8, T, H, W), seed, 0.18215 * 5),
        "x_T": seeding.seeded_input("x_T", (B, 4, T, H, W), seed),
        "noises": [seeding.seeded_input(f"noise{i}", (B, 4, T, H, W), seed) for i in range(s["steps"])],
        "class_label": torch.tensor(s["class_labels"], dtype=torch.long)[:, None],
        "fs": torch.full((B,), s["fs"], dtype=torch.long),
    }


def cached_oracle(key, compute):
    """`compute()` (a CPU-oracle result: a tensor or a tuple / dict of tensors) memoised on disk for the lifetime of the
    box's temp directory.  The same full-size oracle run is needed by the default-mode test and by its re-runs in the
    operand-mode child processes (tests/test_precision_modes_gpu.py) — minutes of host time each; `key` must name
    everything the result depends on (test, sizes, seeds)."""
    import tempfile
    d = os.path.join(tempfile.gettempdir(), "mudg_oracle_cache")
    path = os.path.join(d, key + ".pt")
    if os.path.exists(path):
        try:
            return torch.load(path, map_location="cpu", weights_only=True), True
        except Exception:
            pass
    out = compute()
    try:
        os.makedirs(d, exist_ok=True)
        tmp = path + f".{os.getpid()}.tmp"
        torch.save(out, tmp)
        os.replace(tmp, path)
    except Exception:
        pass
    return out, False


def record_parity(mode, name, value, **extra):
    """Measured parity figures, merged into gpurun_out/parity_modes.json ({mode: {name: value}}): the record bench.py's
    `at_tolerance` quotes once it has been copied to profiles/rN/ (tools/collect_profiles.sh).  The operand-mode children run side
    by side (ChildRuns below), so the read-modify-write is under an advisory file lock."""
    import fcntl
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "parity_modes.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            rec = {}
            if os.path.exists(path):
                with open(path) as f:
                    rec = json.load(f)
            rec.setdefault(mode, {})[name] = dict(extra, value=float(value)) if extra else float(value)
            tmp = path + f".{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                json.dump(rec, f, indent=1, sort_keys=True)
            os.replace(tmp, path)
    except Exception:
        pass


class ChildRuns:
    """Child pytest processes of one test module, up to `workers` of them side by side on the same GPU.  The variant / operand-mode
    children used to run one after the other — two thirds of the suite's 16 minutes, most of it host time (interpreter and torch
    start-up, Python dispatch of small launches, reference arithmetic on the fixtures, model construction) with the GPU idle;
    side by side a block costs about what its longest children do.  Nothing in the suites asserts a time, and every child has its own
    library instance, streams and allocator.  All children are submitted when the first test of the module asks for its result;
    MUDG_CHILDREN_SERIAL=1 runs them one at a time (to bisect a failure that only shows under contention)."""

    def __init__(self, workers):
        from concurrent.futures import ThreadPoolExecutor
        self.workers = 1 if os.environ.get("MUDG_CHILDREN_SERIAL") == "1" else workers
        self.pool = ThreadPoolExecutor(max_workers=self.workers)
        self.futures, self.procs, self.cancelled = {}, [], False
        # The children's reference arithmetic is torch on the host: each child gets its share of the cores.  (Six children with the
        # default of one OpenMP thread per core each — 768 spinning threads on 128 cores — ran the variant block FORTY times slower
        # than one after the other: round-6 log, gpurun_out/r6/gpu_tests_1.txt.)
        self.threads = max(2, (os.cpu_count() or 8) // self.workers)

    def submit(self, key, cmd, cwd, env, timeout, alone=False):
        """alone=True: nothing else starts before this child has finished — the first child of a module runs so, and leaves warm
        on-disk caches (MIOpen's kernel database for the torch references, compiled-kernel caches) to the ones that follow instead
        of having all of them build and lock the same files at once."""
        import subprocess
        import time

        def run():
            if self.cancelled:
                return -999, "cancelled"
            child_env = dict(env)
            if self.workers > 1:
                child_env.update(OMP_NUM_THREADS=str(self.threads), MKL_NUM_THREADS=str(self.threads), OMP_WAIT_POLICY="PASSIVE")
            t0 = time.time()
            proc = subprocess.Popen(cmd, cwd=cwd, env=child_env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            self.procs.append(proc)
            try:
                out, _ = proc.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                proc.kill()
                out, _ = proc.communicate()
                return -998, out + f"\n[timed out after {timeout} s]"
            return proc.returncode, out + f"\n[child {key}: {time.time() - t0:.0f} s wall, started {t0 - self.t0:.0f} s after the first]"

        if key in self.futures:
            return
        if not self.futures:
            self.t0 = time.time()
        if alone:
            fut = self.pool.submit(run)
            self.barrier = fut
            self.futures[key] = fut
        else:
            gate = getattr(self, "barrier", None)

            def gated():
                if gate is not None:
                    gate.result()
                return run()
            self.futures[key] = self.pool.submit(gated)

    def result(self, key):
        return self.futures[key].result()

    def shutdown(self):
        """A failed sibling under -x: do not leave the others running on the GPU."""
        self.cancelled = True
        for proc in self.procs:
            if proc.poll() is None:
                proc.kill()
        self.pool.shutdown(wait=True)
