#!/usr/bin/env python3
"""GroupNorm / LayerNorm at the MDM1024 shapes, one shape per process argument (for kernel-trace passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit

which = sys.argv[1] if len(sys.argv) > 1 else "all"
for (rows, c, samples) in [(294912, 320, 32), (73728, 640, 32), (18432, 1280, 32), (294912, 320, 2)]:
    for dt in (torch.float32, ops.H16()):
        x = torch.randn(rows, c, device="cuda").to(dt)
        g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        sec = timeit(lambda: ops.groupnorm(x, g, b, samples=samples, rows=rows // samples, eps=1e-5, silu=True), iters=10)
        by = rows * c * (2 * x.element_size() + 2)
        print(f"groupnorm rows={rows} C={c} samples={samples} {str(dt)[6:]}: {sec*1e6:7.1f} us {by/sec/1e9:7.0f} GB/s", flush=True)
    x = torch.randn(rows, c, device="cuda")
    sec = timeit(lambda: ops.layernorm(x, g, b), iters=10)
    print(f"layernorm rows={rows} C={c}: {sec*1e6:7.1f} us {rows*c*6/sec/1e9:7.0f} GB/s", flush=True)
