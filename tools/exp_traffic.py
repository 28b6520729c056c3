#!/usr/bin/env python3
"""HBM-side read traffic of individual contraction shapes: run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -- python tools/exp_traffic.py
and summarise with `python tools/rocprof_summary.py pmcd <db> gemm`: one dispatch per shape, in the order printed here
(each shape runs twice; the SECOND dispatch of a pair is the one to read — weights packed, allocator warm)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mudg_amd import ops

dev = torch.device("cuda")
H = ops.H16()
S = ops.STREAM()
SHAPES = [  # (M, N, K, geglu, residual, out_stream)
    (294912, 320, 320, False, True, True), (294912, 320, 1280, False, True, False), (294912, 2560, 320, True, False, False),
    (294912, 960, 320, False, False, False), (73728, 640, 2560, False, True, False), (73728, 5120, 640, True, False, False),
    (18432, 1280, 5120, False, True, False), (18432, 10240, 1280, True, False, False),
]
for (M, N, K, geglu, res, st) in SHAPES:
    x = torch.randn(M, K, device=dev).to(H)
    w = (torch.randn(N, K, device=dev) * 0.05).to(H)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N // 2 if geglu else N, device=dev).to(S) if res else None
    for _ in range(2):
        ops.gemm(x, w, bias=b, geglu=geglu, residual=r, out_stream=st)
    torch.cuda.synchronize()
    nout = N // 2 if geglu else N
    alg = (M * K + N * K) * 2 + (M * nout * 2 if res else 0)
    print(f"M={M} N={N} K={K} geglu={geglu} residual={res}: algorithmic read {alg / 1e6:.0f} MB, write {M * nout * 2 / 1e6:.0f} MB", flush=True)
