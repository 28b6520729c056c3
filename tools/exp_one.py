#!/usr/bin/env python3
"""Run a few GEMM / conv shapes a handful of times each (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import rn

shapes = [(73728, 640, 2560), (18432, 10240, 1280), (294912, 320, 320), (294912, 2560, 320)]
for (M, N, K) in shapes:
    x, w = rn(M, K), rn(N, K)
    for _ in range(3):
        ops.gemm(x, w)
    torch.cuda.synchronize()
