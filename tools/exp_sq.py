#!/usr/bin/env python3
"""Calibration against the guide's GEMM ladder: square bf16 GEMMs, bf16 out, no epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn

for n in (2048, 4096, 8192):
    x, w = rn(n, n), rn(n, n)
    sec = timeit(lambda: ops.gemm(x, w), iters=20)
    print(f"MUDG_GEMM256={os.environ.get('MUDG_GEMM256')} {n}^3: {sec*1e6:8.1f} us {2.0*n**3/sec/1e12:7.1f} TF", flush=True)
