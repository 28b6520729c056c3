#!/usr/bin/env python3
"""Per-shape timing of the MDM1024 (cond+uncond batched) contraction shapes under the current kernel-selection
environment; run several times with MUDG_GEMM256 / MUDG_GEMM256P / MUDG_GEMM_SB set to compare kernels per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn

tag = os.environ.get("TAG", "")
G = [(18432, 10240, 1280, 1), (73728, 5120, 640, 1), (294912, 2560, 320, 1), (18432, 1280, 5120, 0), (18432, 3840, 1280, 0),
     (73728, 1920, 640, 0), (18432, 2560, 1280, 0), (18432, 1280, 1280, 0), (73728, 640, 2560, 0), (294912, 960, 320, 0),
     (4608, 1280, 5120, 0), (4608, 10240, 1280, 1), (4608, 3840, 1280, 0)]
for (M, N, K, g) in G:
    x, w = rn(M, K), rn(N, K)
    b = torch.randn(N, device="cuda")
    sec = timeit(lambda: ops.gemm(x, w, bias=b, geglu=bool(g)), iters=10)
    print(f"{tag} gemm {M} {N} {K} geglu={g}: {sec*1e6:8.1f} us {2.0*M*N*K/sec/1e12:7.1f} TF", flush=True)
T = [(2, 16, 576, 1280), (2, 16, 2304, 640), (2, 16, 9216, 320), (2, 16, 144, 1280)]
for (clips, t, hw, c) in T:
    x, w = rn(clips * t * hw, c), rn(c, 3 * c)
    sec = timeit(lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c), iters=10)
    M = clips * t * hw
    print(f"{tag} tconv {M} {c} {3*c}: {sec*1e6:8.1f} us {2.0*M*c*3*c/sec/1e12:7.1f} TF", flush=True)
C = [(32, 18, 32, 1280, 1280), (32, 18, 32, 2560, 1280), (32, 36, 64, 640, 640), (32, 36, 64, 1920, 640), (32, 72, 128, 320, 320),
     (32, 72, 128, 960, 320), (32, 9, 16, 1280, 1280), (32, 9, 16, 2560, 1280)]
for (f, h, w_, cin, cout) in C:
    x, w = rn(f * h * w_, cin), rn(cout, 9 * cin)
    sec = timeit(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=w_, cin=cin, korder=1), iters=10)
    M = f * h * w_
    print(f"{tag} conv {M} {cout} {9*cin}: {sec*1e6:8.1f} us {2.0*M*cout*9*cin/sec/1e12:7.1f} TF", flush=True)
