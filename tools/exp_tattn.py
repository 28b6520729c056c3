#!/usr/bin/env python3
"""Temporal self-attention at the MDM1024 shapes (cond+uncond batched): time and effective bandwidth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn

for (hw, c) in [(9216, 320), (2304, 640), (576, 1280), (9216, 512)]:
    clips, t = 2, 16
    rows = clips * t * hw
    qkv = rn(rows, 3 * c)
    out = torch.empty(rows, c, device="cuda", dtype=ops.H16())
    sec = timeit(lambda: ops.temporal_attention(qkv, out, clips=clips, t=t, hw=hw, heads=c // 64), iters=10)
    print(f"tattn rows={rows} C={c}: {sec*1e6:8.1f} us {rows*c*8/sec/1e9:7.0f} GB/s", flush=True)
