#!/usr/bin/env python3
"""Micro-benchmarks of the kernels at the shapes the MDM1024 UNet issues (B = 1): TFLOP/s or GB/s per launch.
Run on the GPU box:  python tools/kernel_bench.py [filter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mudg_amd import ops

dev = torch.device("cuda")
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def rn(*s, dtype=BF):
    return (torch.randn(*s, device=dev) * 0.5).to(dtype)


def report(name, sec, flops=None, bytes_=None):
    msg = f"{name:58s} {sec * 1e6:9.1f} us"
    if flops:
        msg += f"  {flops / sec / 1e12:8.1f} TFLOP/s"
    if bytes_:
        msg += f"  {bytes_ / sec / 1e9:8.1f} GB/s"
    print(msg, flush=True)


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    T = 16
    levels = [(72 * 128, 320), (36 * 64, 640), (18 * 32, 1280), (9 * 16, 1280)]
    if "gemm" in flt or not flt:
        for hw, c in levels[:3]:
            M = T * hw
            for (n, k, tag) in [(c, c, "proj CxC"), (2 * c, c, "qk 2CxC"), (3 * c, c, "qkv 3CxC"), (c, 4 * c, "ff2 Cx4C")]:
                x, w = rn(M, k), rn(n, k)
                res = torch.randn(M, n, device=dev)
                report(f"gemm M={M} N={n} K={k} {tag} (+fp32 res/out)", timeit(lambda: ops.gemm(x, w, residual=res, out_fp32=True)), 2.0 * M * n * k)
                report(f"gemm M={M} N={n} K={k} {tag} (bf16 out)", timeit(lambda: ops.gemm(x, w)), 2.0 * M * n * k)
            x, w = rn(M, c), rn(8 * c, c)
            b = torch.randn(8 * c, device=dev)
            report(f"gemm M={M} N={8 * c} K={c} ff1 GEGLU", timeit(lambda: ops.gemm(x, w, bias=b, geglu=True)), 2.0 * M * 8 * c * c)
    if "conv" in flt or not flt:
        for (h, w_, cin, cout) in [(72, 128, 320, 320), (72, 128, 640, 320), (72, 128, 960, 320), (36, 64, 640, 640),
                                   (36, 64, 1280, 640), (18, 32, 1280, 1280), (18, 32, 2560, 1280), (9, 16, 2560, 1280)]:
            x, wt = rn(T * h * w_, cin), rn(cout, 9 * cin)
            for ko in (0, 1):
                report(f"conv3x3 {h}x{w_} {cin}->{cout} korder={ko}", timeit(lambda: ops.conv3x3(x, wt, frames=T, hin=h, win=w_, cin=cin, korder=ko)),
                       2.0 * T * h * w_ * cout * 9 * cin)
        for hw, c in levels:
            x, wt = rn(T * hw, c), rn(c, 3 * c)
            report(f"tconv3 hw={hw} C={c}", timeit(lambda: ops.tconv3(x, wt, clips=1, t=T, hw=hw, cin=c)), 2.0 * T * hw * c * 3 * c)
    if "attn" in flt or not flt:
        for hw, c in levels[:3]:
            heads = c // 64
            qk = rn(T * hw, 2 * c)
            vt = rn(T * c, hw)
            out = torch.empty(T * hw, c, device=dev, dtype=BF)
            report(f"attention N={hw} heads={heads} F={T}", timeit(lambda: ops.attention(qk[:, :c], qk[:, c:], vt, out, frames=T, heads=heads, nq=hw, nk=hw, ldvt=hw, svt=c * hw)),
                   4.0 * T * heads * hw * hw * 64)
            q = rn(T * hw, c); kt = rn(77, c); vtt = rn(c, 80)
            report(f"cross-attn text N={hw} heads={heads}", timeit(lambda: ops.attention(q, kt, vtt, out, frames=T, heads=heads, nq=hw, nk=77, ldvt=80, svt=c * 80, kv_div=T)),
                   4.0 * T * heads * hw * 77 * 64, bytes_=2.0 * T * hw * c * 2)
            qkv = rn(T * hw, 3 * c)
            report(f"temporal attn hw={hw} heads={heads}", timeit(lambda: ops.temporal_attention(qkv, out, clips=1, t=T, hw=hw, heads=heads)),
                   bytes_=T * hw * c * 2.0 * 4)
    if "norm" in flt or not flt:
        for hw, c in [(72 * 128, 320), (72 * 128, 960), (36 * 64, 640), (18 * 32, 1280), (9 * 16, 2560)]:
            x = torch.randn(T * hw, c, device=dev)
            g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
            report(f"groupnorm fp32-in frame-stats hw={hw} C={c}", timeit(lambda: ops.groupnorm(x, g, b, samples=T, rows=hw, eps=1e-5, silu=True)), bytes_=T * hw * c * 10.0)
            report(f"groupnorm fp32-in clip-stats  hw={hw} C={c}", timeit(lambda: ops.groupnorm(x, g, b, samples=1, rows=T * hw, eps=1e-5, silu=True)), bytes_=T * hw * c * 10.0)
            xb = x.to(BF)
            report(f"groupnorm bf16-in clip-stats  hw={hw} C={c}", timeit(lambda: ops.groupnorm(xb, g, b, samples=1, rows=T * hw, eps=1e-5, silu=True)), bytes_=T * hw * c * 6.0)
            report(f"layernorm fp32-in rows={T * hw} C={c}", timeit(lambda: ops.layernorm(x, g, b)), bytes_=T * hw * c * 6.0)


if __name__ == "__main__":
    main()
