#!/usr/bin/env python3
"""A bounded experiment on "fewer MFMAs per product" in the bf16x3 mode (round-5 review, item 2c), done as NUMERICAL EMULATION on the
CPU before any kernel is written: would the two low-order cross terms of a bf16x3 product — x1 w0 and x0 w1, which carry 2^-8 of the
sum — survive being computed from MX-fp8 pieces (e4m3 + one E8M0 scale per 32 values along K, the operand format of
v_mfma_scale_f32_32x32x64_f8f6f4, twice the bf16 MFMA rate: 2 instead of 3 MFMA-equivalents per product)?

Every contraction of the oracle's UNet forward (oracle/unet.py: _lin, _conv — Linear, 1x1 / 3x3 / temporal convs; the attention matmuls
are left exact in every mode) is replaced by an emulation of the operand mode:
    bf16        x0 w0                                   (operands rounded to bf16, exact accumulation)
    bf16x3      x0 w0 + x1 w0 + x0 w1                   (x = x0 + x1, w = w0 + w1, pieces bf16)
    x3-fp8cross x0 w0 + q8(x1) q8(w0) + q8(x0) q8(w1)   (q8 = MX-fp8 of the piece, blocks of 32 along the channel axis)
and the small golden fixture (tests/golden/unet_a.pt: the reference's own topology at reduced width) is run in each; the per-forward
rel-L2 against the fp32 oracle is printed, next to the measured GPU figures of the real kernels for scale.

    python tools/exp_cross_fp8.py            (CPU, about a minute)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from helpers import golden, rel_l2, seeded_sd, unet_inputs
from oracle import unet as o_unet


def bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


def q8(x, dim):
    """MX-fp8: blocks of 32 along `dim`, shared exponent E = floor(log2(amax)) - 8 (csrc/common.h: mx_block_exponent), values e4m3."""
    x = x.movedim(dim, -1)
    shp = x.shape
    k = shp[-1]
    pad = (-k) % 32
    xp = F.pad(x, (0, pad)).reshape(*shp[:-1], -1, 32)
    amax = xp.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(1e-38))) - 8
    e = torch.where(amax > 0, e, torch.zeros_like(e)).clamp(-127, 127)
    s = torch.exp2(e)
    q = (xp / s).clamp(-448, 448).to(torch.float8_e4m3fn).to(x.dtype) * s
    return q.reshape(*shp[:-1], -1)[..., :k].movedim(-1, dim)


MODE = {"name": "fp32"}


def contract(fn, x, w, xdim, wdim):
    """fn(x, w) under the emulated operand mode; xdim / wdim = the contracted (channel) axis of each operand."""
    m = MODE["name"]
    if m == "fp32":
        return fn(x, w)
    x, w = x.double(), w.double()
    x0, w0 = bf(x), bf(w)
    if m == "bf16":
        return fn(x0, w0).float()
    x1, w1 = bf(x - x0), bf(w - w0)
    if m == "bf16x3":
        return (fn(x0, w0) + fn(x1, w0) + fn(x0, w1)).float()
    if m == "x3-fp8cross":
        return (fn(x0, w0) + fn(q8(x1, xdim), q8(w0, wdim)) + fn(q8(x0, xdim), q8(w1, wdim))).float()
    if m == "x3-fp8cross-w1only":      # only the x0 w1 term from fp8 pieces (x1 w0 stays bf16): 2.5 MFMA-equivalents
        return (fn(x0, w0) + fn(x1, w0) + fn(q8(x0, xdim), q8(w1, wdim))).float()
    raise ValueError(m)


def _lin(sd, p, x):
    b = sd.get(p + ".bias")
    y = contract(lambda a, w: F.linear(a, w), x, sd[p + ".weight"], -1, -1)
    return y if b is None else y + b


def _conv(sd, p, x, **kw):
    w = sd[p + ".weight"]
    fn = {3: F.conv1d, 4: F.conv2d, 5: F.conv3d}[w.dim()]
    b = sd.get(p + ".bias")
    y = contract(lambda a, ww: fn(a, ww, None, **kw), x, w, 1, 1)
    return y if b is None else y + b.reshape(1, -1, *([1] * (w.dim() - 2)))


def main():
    o_unet._lin, o_unet._conv = _lin, _conv
    torch.manual_seed(0)
    # one GEMM first: the per-product error of each mode
    x, w = torch.randn(2048, 1280), torch.randn(640, 1280) * 0.02
    ref = F.linear(x.double(), w.double())
    print("one 2048 x 640 x 1280 product, rel-L2 against the exact result:")
    per_gemm = {}
    for m in ("bf16", "bf16x3", "x3-fp8cross-w1only", "x3-fp8cross"):
        MODE["name"] = m
        per_gemm[m] = rel_l2(contract(lambda a, b: F.linear(a, b), x, w, -1, -1), ref)
        print(f"  {m:20s} {per_gemm[m]:.3e}")
    g = golden("unet_a.pt")
    sd = seeded_sd(g["param_shapes"], g["seed"], g["checksum"])
    xin, ctx = unet_inputs(g["cfg"], g["shape"], g["seed"])
    case = g["cases"][0]
    MODE["name"] = "fp32"
    want = o_unet.unet_forward(sd, g["cfg"], xin, case["t"], case["c_label"], ctx, case["fs"])
    print(f"UNet forward of the golden fixture (tests/golden/unet_a.pt), every Linear / conv emulated, rel-L2 against the fp32 oracle "
          f"(which is at {rel_l2(want, case['y']):.1e} of the reference's own output):")
    per_fwd = {}
    for m in ("bf16", "bf16x3", "x3-fp8cross-w1only", "x3-fp8cross"):
        MODE["name"] = m
        per_fwd[m] = rel_l2(o_unet.unet_forward(sd, g["cfg"], xin, case["t"], case["c_label"], ctx, case["fs"]), want)
        print(f"  {m:20s} {per_fwd[m]:.3e}")
    # scale: the real kernels at MDM1024 (profiles/r5/parity_modes.json): one guided step + one decoded 576 x 1024 frame
    bf16_e2e, x3_e2e = 1.30e-1, 2.7e-4
    amp = x3_e2e / per_fwd["bf16x3"]
    print(f"scale: the real bf16x3 kernels measure {x3_e2e:.1e} on the MDM1024 guided step + decoded frame (bf16: {bf16_e2e:.1e}); the decoded-frame "
          f"error follows the per-forward error of the contractions linearly (bf16 / bf16x3: emulated ratio {per_fwd['bf16'] / per_fwd['bf16x3']:.0f}, "
          f"measured ratio {bf16_e2e / x3_e2e:.0f}).")
    for m in ("x3-fp8cross-w1only", "x3-fp8cross"):
        est = per_fwd[m] * amp
        print(f"  {m:20s} -> estimated {est:.1e} on that test (contract 1e-3, keep-threshold of the review 5e-4): "
              f"{'inside' if est <= 5e-4 else 'OUTSIDE'}")


if __name__ == "__main__":
    main()
