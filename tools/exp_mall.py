#!/usr/bin/env python3
"""Does a consumer that runs right after its producer read from the 256 MB Infinity Cache faster than from HBM?
LayerNorm on an fp32 tensor of S MB, (a) right after a kernel wrote that tensor, (b) after 1 GB of unrelated traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops

dev = "cuda"
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB
c = 320
for mb in (24, 47, 94, 189, 377):
    rows = mb * (1 << 20) // (c * 4)
    src = torch.randn(rows, c, device=dev)
    x = torch.empty_like(src)
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    out = ops.empty_rows(rows, c)
    res = {}
    for mode in ("warm", "cold"):
        ts = []
        for it in range(6):
            x.copy_(src)                          # producer writes x
            if mode == "cold":
                flush.add_(1.0)                   # 2 GiB of unrelated traffic evicts it
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.layernorm(x, g, b, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = sorted(ts)[len(ts) // 2] * 1e3
    by = rows * c * 6
    print(f"LayerNorm fp32 {mb} MB: warm {res['warm']:7.1f} us ({by/res['warm']/1e3:6.0f} GB/s)  cold {res['cold']:7.1f} us ({by/res['cold']/1e3:6.0f} GB/s)", flush=True)
