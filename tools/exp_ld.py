#!/usr/bin/env python3
"""Experiment: does the row stride of X / W (L2 channel spread) change GEMM throughput?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn

for (M, N, K) in [(18432, 1280, 5120), (18432, 3840, 1280), (73728, 640, 2560), (294912, 320, 1280), (18432, 10240, 1280), (73728, 1920, 640)]:
    for padx, padw in [(0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (8, 8)]:
        xb, wb = rn(M, K + padx), rn(N, K + padw)
        x, w = xb[:, :K], wb[:, :K]
        sec = timeit(lambda: ops.gemm(x, w, K=K), iters=20)
        print(f"M={M} N={N} K={K} ldx=K+{padx} ldw=K+{padw}: {sec*1e6:8.1f} us {2.0*M*N*K/sec/1e12:7.1f} TF", flush=True)
