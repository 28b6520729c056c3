#!/usr/bin/env python3
"""Which op, if any, gives different bits for the same rows when the batch around them changes?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudg_amd import ops
from tools.kernel_bench import rn

def report(name, full, parts):
    cat = torch.cat(parts, 0)
    same = torch.equal(full, cat)
    d = (full.float() - cat.float()).norm() / full.float().norm()
    print(f"{name:50s} bit-identical={same} rel={d.item():.2e}", flush=True)

T, hw, c = 16, 9216, 320
rows = 2 * T * hw
x = rn(rows, c); half = rows // 2
w = rn(960, c); b = torch.randn(960, device="cuda")
report("gemm N=960 K=320", ops.gemm(x, w, bias=b), [ops.gemm(x[:half], w, bias=b), ops.gemm(x[half:], w, bias=b)])
wg = rn(2560, c)
report("gemm geglu N=2560", ops.gemm(x, wg, geglu=True), [ops.gemm(x[:half], wg, geglu=True), ops.gemm(x[half:], wg, geglu=True)])
res = torch.randn(rows, c, device="cuda"); w2 = rn(c, c)
report("gemm N=320 res32 out32", ops.gemm(x, w2, residual=res, out_fp32=True), [ops.gemm(x[:half], w2, residual=res[:half], out_fp32=True), ops.gemm(x[half:], w2, residual=res[half:], out_fp32=True)])
wc = rn(c, 9 * c)
f = lambda xx, fr: ops.conv3x3(xx, wc, frames=fr, hin=72, win=128, cin=c, korder=1, stats=True)
cf = f(x, 32); c1 = f(x[:half], 16); c2 = f(x[half:], 16)
report("conv3x3 320->320 ds1", cf, [c1, c2])
wt = rn(c, 3 * c)
report("tconv3 ds1", ops.tconv3(x, wt, clips=2, t=T, hw=hw, cin=c), [ops.tconv3(x[:half], wt, clips=1, t=T, hw=hw, cin=c), ops.tconv3(x[half:], wt, clips=1, t=T, hw=hw, cin=c)])
g, bb = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
report("groupnorm fused (spatial)", ops.groupnorm(cf, g, bb, samples=32, rows=hw, eps=1e-5, silu=True), [ops.groupnorm(c1, g, bb, samples=16, rows=hw, eps=1e-5, silu=True), ops.groupnorm(c2, g, bb, samples=16, rows=hw, eps=1e-5, silu=True)])
report("groupnorm 2-pass (spatial)", ops.groupnorm(x, g, bb, samples=32, rows=hw, eps=1e-5, silu=True, fused=False), [ops.groupnorm(x[:half], g, bb, samples=16, rows=hw, eps=1e-5, silu=True, fused=False), ops.groupnorm(x[half:], g, bb, samples=16, rows=hw, eps=1e-5, silu=True, fused=False)])
report("groupnorm 2-pass (temporal)", ops.groupnorm(x, g, bb, samples=2, rows=T * hw, eps=1e-5, silu=True, fused=False), [ops.groupnorm(x[:half], g, bb, samples=1, rows=T * hw, eps=1e-5, silu=True, fused=False), ops.groupnorm(x[half:], g, bb, samples=1, rows=T * hw, eps=1e-5, silu=True, fused=False)])
xf = torch.randn(rows, c, device="cuda")
report("layernorm", ops.layernorm(xf, g, bb), [ops.layernorm(xf[:half], g, bb), ops.layernorm(xf[half:], g, bb)])
qkv = rn(rows, 3 * c)
def ta(q, clips):
    o = torch.empty(q.shape[0], c, device="cuda", dtype=ops.H16()); ops.temporal_attention(q, o, clips=clips, t=T, hw=hw, heads=5); return o
report("temporal attention", ta(qkv, 2), [ta(qkv[:half].contiguous(), 1), ta(qkv[half:].contiguous(), 1)])
def sa(frames, qk, vt):
    o = torch.empty(frames * hw, c, device="cuda", dtype=ops.H16())
    ops.attention(qk[:, :c], qk[:, c:], vt, o, frames=frames, heads=5, nq=hw, nk=hw, ldvt=hw, svt=c * hw); return o
qk = rn(32 * hw, 2 * c); vt = rn(32 * c, hw)
report("spatial attention", sa(32, qk, vt), [sa(16, qk[:16 * hw].contiguous(), vt[:16 * c].contiguous()), sa(16, qk[16 * hw:].contiguous(), vt[16 * c:].contiguous())])
# ds8-like shapes (rows per frame 144: two-pass GroupNorm, few tiles)
x8 = rn(32 * 144, 1280); w8 = rn(1280, 9 * 1280); h8 = 16 * 144
f8 = lambda xx, fr: ops.conv3x3(xx, w8, frames=fr, hin=9, win=16, cin=1280, korder=1, stats=True)
report("conv3x3 1280 ds8", f8(x8, 32), [f8(x8[:h8], 16), f8(x8[h8:], 16)])
g8, b8 = torch.ones(1280, device="cuda"), torch.zeros(1280, device="cuda")
report("groupnorm ds8 (spatial, 2-pass)", ops.groupnorm(x8, g8, b8, samples=32, rows=144, eps=1e-5, silu=True), [ops.groupnorm(x8[:h8], g8, b8, samples=16, rows=144, eps=1e-5, silu=True), ops.groupnorm(x8[h8:], g8, b8, samples=16, rows=144, eps=1e-5, silu=True)])
