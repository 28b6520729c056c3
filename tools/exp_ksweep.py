#!/usr/bin/env python3
"""K sweep at fixed M, N: time = fixed per-tile cost + per-K-tile cost (short-K analysis)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn

M = 294912
for (N, geglu) in [(2560, True), (2560, False), (320, False)]:
    for K in (64, 128, 320, 640, 1280):
        x, w = rn(M, K), rn(N, K)
        b = torch.randn(N, device="cuda")
        sec = timeit(lambda: ops.gemm(x, w, bias=b, geglu=geglu), iters=10)
        tiles = (M // 128) * ((N + 127) // 128)
        print(f"M={M} N={N} K={K} geglu={geglu}: {sec*1e6:8.1f} us {2.0*M*N*K/sec/1e12:7.1f} TF  {sec*1e9/tiles:6.2f} ns/tile", flush=True)
