#!/usr/bin/env python3
"""BASELINE config 5 ("fp8 MFMA attention"): what P.V on the fp8 MFMA would cost in accuracy, by NUMERICAL EMULATION on the CPU (the
review's item 8 asked for P.V in fp8 with the error next to the Q.K^T-only figure, or the measured reason not to).

The oracle's UNet forward on the golden fixture (tests/golden/unet_a.pt) with every Linear / conv on bf16 operands (the benchmarked
mode) and the spatial SELF-attention emulated three ways:
    bf16             q, k, v and P rounded to bf16                              (the default build)
    fp8 scores       q, k as MX-fp8 (e4m3 + E8M0 per 32 along the head axis), P and v bf16      (MUDG_ATTN_FP8=1 as shipped)
    fp8 scores + PV  additionally P and v as MX-fp8 along the KEY axis (the contraction axis of P.V; P in [0, 1] after the running-max
                     subtraction — the friendly case for e4m3)
    python tools/exp_fp8_pv.py       (CPU, about a minute)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torch.nn.functional as F
from helpers import golden, rel_l2, seeded_sd, unet_inputs
from oracle import unet as o_unet
import exp_cross_fp8 as X

ATT = {"mode": "exact"}
bf = X.bf
q8 = X.q8


def attention_core(q, k, v, heads, head_chunk=None):
    b, n, c = q.shape
    d = c // heads
    self_attn = k.shape[1] == n and n >= 64
    q = q.reshape(b, n, heads, d).transpose(1, 2).double()
    k = k.reshape(b, k.shape[1], heads, d).transpose(1, 2).double()
    v = v.reshape(b, v.shape[1], heads, d).transpose(1, 2).double()
    m = ATT["mode"]
    if m == "exact":
        p = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
        out = p @ v
    else:
        f8 = self_attn and m != "bf16"
        qq, kk = (q8(bf(q), -1), q8(bf(k), -1)) if f8 else (bf(q), bf(k))
        s = qq @ kk.transpose(-1, -2) * d ** -0.5
        p = torch.exp(s - s.amax(-1, keepdim=True))                  # un-normalised, in (0, 1]: what the kernel multiplies with V
        l = p.sum(-1, keepdim=True)
        if self_attn and m == "fp8pv":
            out = (q8(p, -1) @ q8(bf(v), -2)) / l
        else:
            out = (bf(p) @ bf(v)) / l
    return out.float().transpose(1, 2).reshape(b, n, c)


def main():
    o_unet._lin, o_unet._conv, o_unet.attention_core = X._lin, X._conv, attention_core
    g = golden("unet_a.pt")
    sd = seeded_sd(g["param_shapes"], g["seed"], g["checksum"])
    xin, ctx = unet_inputs(g["cfg"], g["shape"], g["seed"])
    case = g["cases"][0]
    X.MODE["name"], ATT["mode"] = "fp32", "exact"
    want = o_unet.unet_forward(sd, g["cfg"], xin, case["t"], case["c_label"], ctx, case["fs"])
    print("UNet forward of the golden fixture, Linear / conv on bf16 operands, rel-L2 against the fp32 oracle:")
    res = {}
    X.MODE["name"] = "bf16"
    for m, name in (("bf16", "bf16 attention"), ("fp8s", "MX-fp8 scores (shipped config 5)"), ("fp8pv", "MX-fp8 scores + MX-fp8 P.V")):
        ATT["mode"] = m
        res[m] = rel_l2(o_unet.unet_forward(sd, g["cfg"], xin, case["t"], case["c_label"], ctx, case["fs"]), want)
        print(f"  {name:36s} {res[m]:.3e}  (x {res[m] / res['bf16']:.2f} of bf16)")
    # attention alone on one long layer-like problem
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, 2304, 320) for _ in range(3))
    ATT["mode"] = "exact"
    ref = attention_core(q, k, v, 5)
    print("one self-attention (2304 tokens, 5 heads of 64), rel-L2 against exact:")
    for m in ("bf16", "fp8s", "fp8pv"):
        ATT["mode"] = m
        print(f"  {m:8s} {rel_l2(attention_core(q, k, v, 5), ref):.3e}")
    print("measured with the real kernels at MDM1024 (profiles/r5/parity_modes.json): decoded frame bf16 1.30e-1, MX-fp8 scores 1.48e-1 (x 1.14).")


if __name__ == "__main__":
    main()
