#!/usr/bin/env python3
"""Calibration against the guide's GEMM ladder: square bf16 GEMMs, bf16 out, no epilogue.  PAD=n pads both row strides."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn

pad = int(os.environ.get("PAD", "0"))
for n in (2048, 4096, 8192):
    xb, wb = rn(n, n + pad), rn(n, n + pad)
    x, w = xb[:, :n], wb[:, :n]
    sec = timeit(lambda: ops.gemm(x, w, K=n), iters=20)
    print(f"MUDG_GEMM256={os.environ.get('MUDG_GEMM256')} P={os.environ.get('MUDG_GEMM256P')} pad={pad} {n}^3: {sec*1e6:8.1f} us {2.0*n**3/sec/1e12:7.1f} TF", flush=True)
