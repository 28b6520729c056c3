#!/usr/bin/env python3
"""Calibration: PyTorch's bf16 matmul (hipBLASLt / rocBLAS) on the plain-GEMM shapes of the path next to mudg_gemm
without epilogue — a same-hardware reference for what these shapes can reach (not used by the product path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn

for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (294912, 2560, 320), (294912, 320, 320), (294912, 960, 320), (294912, 320, 1280),
                  (73728, 5120, 640), (73728, 640, 2560), (73728, 1920, 640), (18432, 10240, 1280), (18432, 1280, 5120), (18432, 3840, 1280)]:
    x, w = rn(M, K), rn(N, K)
    wt = w.t().contiguous()
    t_lib = timeit(lambda: torch.matmul(x, wt), iters=10)
    t_lib2 = timeit(lambda: torch.nn.functional.linear(x, w), iters=10)
    t_own = timeit(lambda: ops.gemm(x, w), iters=10)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: torch.matmul {fl/t_lib/1e12:7.1f} TF | F.linear {fl/t_lib2/1e12:7.1f} TF | mudg_gemm {fl/t_own/1e12:7.1f} TF", flush=True)
