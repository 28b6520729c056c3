#!/usr/bin/env python3
"""The half-height two-workgroup GEGLU kernel (csrc/wgemm.hip: hgeglu_kernel, round 6) against the kernels it replaces, on the same inputs:
bit-identity on small and ragged problems, then per-shape timings of the benchmark's GEGLU shapes (MDM1024 and MDM512).
    MUDG_DEBUG_VARIANTS=1 python tools/exp_h144.py [parity|time|all]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MUDG_DEBUG_VARIANTS", "1")
import torch
from mudg_amd import hip, ops
from tools.kernel_bench import timeit

what = sys.argv[1] if len(sys.argv) > 1 else "all"
rn = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(ops.H16())
FORMS = (("128x128", "0", "0"), ("288x256", "2", "0"), ("144x256 x2", "2", "2"))      # (name, GEMM_W288, GEMM_H144)


def setv(w288, h144):
    os.environ["MUDG_GEMM_W288"], os.environ["MUDG_GEMM_H144"] = w288, h144


if what in ("parity", "all"):
    torch.manual_seed(0)
    for M, N, K, f32 in ((144 * 5, 512, 320, False), (144 * 7 + 13, 256, 64, False), (1, 256, 128, True), (288 * 40 + 100, 2560, 320, False),
                         (144 * 33, 1024, 1280, True), (5000, 768, 192, False), (81920, 2560, 320, False)):
        x, w, b = rn(M, K), rn(N, K) * 0.2, torch.randn(N, device="cuda")
        outs = []
        for name, v, h in FORMS:
            setv(v, h)
            outs.append(ops.gemm(x, w, bias=b, geglu=True, out_fp32=f32, frame_rows=288))
        torch.cuda.synchronize()
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        ref = ref[:, :N // 2].reshape(M, -1)      # packed GEGLU weights interleave [32 value | 32 gate]: compare the kernels with each other only
        print(f"geglu {M}x{N}x{K} fp32={f32}: 144x256 == 288x256: {torch.equal(outs[2], outs[1])}; == 128x128: {torch.equal(outs[2], outs[0])}; "
              f"finite {bool(torch.isfinite(outs[2].float()).all())}", flush=True)
    x2 = rn(144 * 9, 128)
    x, w, b = rn(144 * 9, 192), rn(512, 320) * 0.2, torch.randn(512, device="cuda")
    outs = []
    for name, v, h in FORMS:
        setv(v, h)
        outs.append(ops.gemm(x, w, x2=x2, bias=b, geglu=True, frame_rows=288))
    print(f"geglu two sources: 144x256 == 288x256: {torch.equal(outs[2], outs[1])}; == 128x128: {torch.equal(outs[2], outs[0])}", flush=True)

if what in ("time", "all"):
    for (M, N, K, hw) in [(294912, 2560, 320, 9216), (73728, 5120, 640, 2304), (18432, 10240, 1280, 576), (147456, 4096, 512, 9216), (4608, 10240, 1280, 144),
                          (81920, 2560, 320, 2560), (20480, 5120, 640, 640), (5120, 10240, 1280, 160)]:
        x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
        ts = []
        for name, v, h in (("rule-r5", "1", "0"),) + FORMS[1:]:
            setv(v, h)
            ts.append(timeit(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw), iters=20))
        fl = 2.0 * M * N * K
        print(f"geglu {M} {N} {K}: round-5 rule {ts[0]*1e6:8.1f} us {fl/ts[0]/1e12:7.1f} TF | 288x256 one-tile {ts[1]*1e6:8.1f} us | 144x256 x2 {ts[2]*1e6:8.1f} us "
              f"{fl/ts[2]/1e12:7.1f} TF  x{ts[0]/ts[2]:.3f}", flush=True)
if what == "delay":
    # the anti-phase start of the two workgroups of a CU (GEMM_H144DELAY, units of 8128 cycles)
    for (M, N, K, hw) in [(294912, 2560, 320, 9216), (73728, 5120, 640, 2304), (18432, 10240, 1280, 576), (81920, 2560, 320, 2560)]:
        x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
        line = f"geglu {M} {N} {K}:"
        setv("1", "0")
        line += f" round-5 rule {timeit(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw), iters=20)*1e6:7.1f} us |"
        setv("2", "2")
        for d in (0, 1, 2, 3, 4, 6):
            os.environ["MUDG_GEMM_H144DELAY"] = str(d)
            line += f" delay {d}: {timeit(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw), iters=20)*1e6:7.1f}"
        print(line, flush=True)
if what == "ablate":
    # the two-workgroup kernel with parts switched off (GEMM_H144ABL: 1 = no K loop, 2 = no epilogue, 3 = neither; wrong results, timing only)
    for (M, N, K, hw) in [(294912, 2560, 320, 9216), (73728, 5120, 640, 2304), (18432, 10240, 1280, 576)]:
        x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
        setv("2", "2")
        line = f"geglu {M} {N} {K}:"
        for a, name in ((0, "full"), (1, "no K loop"), (2, "no epilogue"), (3, "launch + table copy only")):
            os.environ["MUDG_GEMM_H144ABL"] = str(a)
            line += f" {name} {timeit(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw), iters=20)*1e6:7.1f} us |"
        os.environ["MUDG_GEMM_H144ABL"] = "0"
        print(line, flush=True)
if what == "prio":
    for (M, N, K, hw) in [(294912, 2560, 320, 9216), (73728, 5120, 640, 2304), (18432, 10240, 1280, 576), (81920, 2560, 320, 2560)]:
        x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
        setv("1", "0")
        line = f"geglu {M} {N} {K}: persistent 288x256 {timeit(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw), iters=20)*1e6:7.1f} us |"
        setv("2", "2")
        for pr in ("0", "1"):
            os.environ["MUDG_GEMM_H144PRIO"] = pr
            line += f" 144x256 x2 prio {pr}: {timeit(lambda: ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw), iters=20)*1e6:7.1f} us |"
        print(line, flush=True)
if what == "one":
    # one launch per form of the level-0 shape, for rocprofv3 --pmc (read the LAST dispatch of each kernel)
    M, N, K, hw = 294912, 2560, 320, 9216
    x, w, b = rn(M, K), rn(N, K), torch.randn(N, device="cuda")
    for name, v, h in (("rule-r5", "1", "0"), ("h144", "2", "2")):
        setv(v, h)
        for _ in range(3):
            ops.gemm(x, w, bias=b, geglu=True, frame_rows=hw)
        torch.cuda.synchronize()
setv("1", "1")
