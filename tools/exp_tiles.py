#!/usr/bin/env python3
"""Per-shape timing of the MDM1024 (cond + uncond batched) contraction shapes under the current kernel-selection
environment.  Run it on the debug-variants build to compare kernels per shape:
    MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_WIDE=0 TAG=g128 python tools/exp_tiles.py
    MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_WIDE=1 TAG=wide python tools/exp_tiles.py
(GEMM_WIDE: 0 = 128 x 128 tiles only, 1 = the 256 x 320 / 256 x 256 tiles wherever N fits, default = the library's rule).
RES=512: the same list at the MDM512 resolution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import timeit, rn

tag = os.environ.get("TAG", "")
only = os.environ.get("ONLY", "")
RESID = os.environ.get("RESID", "1") == "1"          # plain GEMMs with the residual epilogue (the out-projection / FF2 form)
STATS = os.environ.get("STATS", "1") == "1"          # convs with GroupNorm partials
if os.environ.get("RES", "1024") == "512":        # the MDM512 shapes: 40 x 64 latent pixels instead of 72 x 128
    HWS, DIMS = {9216: 2560, 2304: 640, 576: 160, 144: 40}, {72: 40, 128: 64, 36: 20, 64: 32, 18: 10, 32: 16, 9: 5, 16: 8}
else:
    HWS, DIMS = {}, {}
G = [(294912, 2560, 320, 1), (73728, 5120, 640, 1), (18432, 10240, 1280, 1), (294912, 320, 320, 0), (294912, 320, 1280, 0),
     (294912, 960, 320, 0), (73728, 640, 640, 0), (73728, 640, 2560, 0), (73728, 1920, 640, 0), (18432, 1280, 5120, 0),
     (18432, 3840, 1280, 0), (18432, 1280, 1280, 0), (18432, 2560, 1280, 0), (4608, 1280, 5120, 0), (4608, 10240, 1280, 1),
     (4608, 3840, 1280, 0)]
if not only or "gemm" in only:
    for (M, N, K, g) in G:
        M = M // 32 and 32 * HWS.get(M // 32, M // 32)
        x, w = rn(M, K), rn(N, K)
        b = torch.randn(N, device="cuda")
        r = rn(M, N // 2 if g else N).to(ops.STREAM())
        sec = timeit(lambda: ops.gemm(x, w, bias=b, geglu=bool(g), residual=None if (g or not RESID) else r, out_stream=not g), iters=10)
        print(f"{tag} gemm {M} {N} {K} geglu={g}: {sec*1e6:8.1f} us {2.0*M*N*K/sec/1e12:7.1f} TF", flush=True)
T = [(2, 16, 9216, 320), (2, 16, 2304, 640), (2, 16, 576, 1280), (2, 16, 144, 1280)]
if not only or "tconv" in only:
    for (clips, t, hw, c) in T:
        hw = HWS.get(hw, hw)
        x, w = rn(clips * t * hw, c), rn(c, 3 * c)
        sec = timeit(lambda: ops.tconv3(x, w, clips=clips, t=t, hw=hw, cin=c, stats=STATS, korder=int(os.environ.get("TKORDER", "0"))), iters=10)
        M = clips * t * hw
        print(f"{tag} tconv {M} {c} {3*c}: {sec*1e6:8.1f} us {2.0*M*c*3*c/sec/1e12:7.1f} TF", flush=True)
C = [(32, 72, 128, 320, 320), (32, 72, 128, 640, 320), (32, 72, 128, 960, 320), (32, 36, 64, 640, 640), (32, 36, 64, 1280, 640),
     (32, 36, 64, 1920, 640), (32, 18, 32, 1280, 1280), (32, 18, 32, 2560, 1280), (32, 9, 16, 1280, 1280), (32, 9, 16, 2560, 1280)]
if not only or "conv3" in only:
    for (f, h, w_, cin, cout) in C:
        h, w_ = (DIMS[h], DIMS[w_]) if DIMS else (h, w_)
        x, w = rn(f * h * w_, cin), rn(cout, 9 * cin)
        sec = timeit(lambda: ops.conv3x3(x, w, frames=f, hin=h, win=w_, cin=cin, korder=1, stats=STATS), iters=10)
        M = f * h * w_
        print(f"{tag} conv {M} {cout} {9*cin}: {sec*1e6:8.1f} us {2.0*M*cout*9*cin/sec/1e12:7.1f} TF", flush=True)
