#!/usr/bin/env python3
"""Spatial self-attention at the MDM1024 ds1 / ds2 shapes (cond+uncond batched): timing, for PMC passes too."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn

for (frames, heads, n) in [(32, 5, 9216), (32, 10, 2304)]:
    C = heads * 64
    qk = rn(frames * n, 2 * C)
    vt = rn(frames * C, n)
    out = torch.empty(frames * n, C, device="cuda", dtype=ops.H16())
    fn = lambda: ops.attention(qk[:, :C], qk[:, C:], vt, out, frames=frames, heads=heads, nq=n, nk=n, ldvt=n, svt=C * n)
    sec = timeit(fn, iters=5, warm=2)
    fl = 4.0 * frames * heads * n * n * 64
    print(f"attention frames={frames} heads={heads} N={n}: {sec*1e6:8.1f} us {fl/sec/1e12:7.1f} TF", flush=True)
