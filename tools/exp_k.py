#!/usr/bin/env python3
"""Fixed cost of an output tile against the cost of a K-step: the level-0 GEMM shapes (M = 294912, N = 2560 / 1280) timed at
K = 64 ... 1280, with and without the GEGLU epilogue.  `python tools/exp_k.py` on an MI355X (results: DESIGN.md §6)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn
M = 294912
for g in (1, 0):
    for N in ((2560,) if g else (2560, 1280)):
        for K in (64, 128, 320, 640, 1280):
            x, w = rn(M, K), rn(N, K)
            b = torch.randn(N, device="cuda")
            sec = timeit(lambda: ops.gemm(x, w, bias=b, geglu=bool(g)), iters=5)
            tiles = (M // 128) * (N // 128)
            print(f"geglu={g} N={N} K={K}: {sec*1e6:8.1f} us  {2.0*M*N*K/sec/1e12:6.1f} TF  per tile per CU {sec*1e6/(tiles/256):.2f} us", flush=True)
