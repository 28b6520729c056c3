"""Kernel parity that holds in EVERY operand mode of the library (bf16, fp16, bf16x3, bf16x6).

Inputs are fp32 values converted to MFMA operands by the library itself (mudg_cast_rows: one 16-bit number, or 2 / 3
bf16 pieces per value in the split-operand builds), read back exactly with the same entry point, and the reference is
computed in fp64 from those read-back values — so what is measured is the kernels' arithmetic, not the input rounding.
Operand outputs carry one operand rounding (2^-9 bf16, 2^-12 fp16, ~2^-18 x3, ~2^-26 x6); fp32 outputs carry what the
mode drops inside the contraction (nothing in the 16-bit modes, the x1*w1 partial product in x3).

This file runs in the default mode directly, and once per other mode in a child process (test_precision_modes_gpu.py)."""
import os

import pytest
import torch
import torch.nn.functional as F

from mudg_amd import hip as _hip

pytestmark = pytest.mark.gpu

MODE = _hip.operand_name()
# (operand-output tolerance, fp32-output tolerance) as rel-L2
TOL_OP, TOL_F32 = {"bf16": (3e-3, 2e-5), "fp16": (4e-4, 2e-5), "bf16x3": (1.5e-5, 1.5e-5), "bf16x6": (1e-6, 1e-6)}[MODE]


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def f32(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def operand(x, cuda):
    """fp32 [rows, C] -> (operand rows on the GPU, the exact values they hold as fp64 on the CPU)."""
    from mudg_amd import ops
    t = ops.cast_bf16(x.to(cuda).float().contiguous())
    return t, ops.to_f32(t).double().cpu()


def value(t):
    from mudg_amd import ops
    return (ops.to_f32(t) if t.dtype != torch.float32 else t).double().cpu()


def test_cast_round_trip_is_exact_to_the_modes_precision(cuda):
    x = f32(300, 72, seed=1)
    t, v = operand(x, cuda)
    assert tuple(t.shape) == (300, 72) and t.stride(0) == 72 * _hip.planes()
    eps = {"bf16": 2 ** -8, "fp16": 2 ** -11, "bf16x3": 2 ** -17, "bf16x6": 2 ** -24}[MODE]
    assert bool(((v - x.double()).abs() <= eps * x.double().abs() + 2.0 ** -25).all())      # + fp16's subnormal spacing
    from mudg_amd import ops
    again = ops.cast_rows(t, ops.empty_rows(300, 72, None, cuda))        # operand -> operand keeps every piece
    assert torch.equal(value(again), v)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 96, 72), (1000, 320, 320), (77, 256, 1024), (300, 4, 320),
                                   (129, 200, 8), (2000, 128, 640)])
def test_gemm(cuda, M, N, K):
    from mudg_amd import ops
    x, xv = operand(f32(M, K, seed=1), cuda)
    w, wv = operand(f32(N, K, seed=2, scale=0.05), cuda)
    r, rv = operand(f32(M, N, seed=4), cuda)
    b = f32(N, seed=3)
    ref = xv @ wv.t() + b.double() + rv
    y = ops.gemm(x, w, bias=b.to(cuda), residual=r)
    assert tuple(y.shape) == (M, N) and rel(value(y), ref) < TOL_OP
    r32 = f32(M, N, seed=5)
    y32 = ops.gemm(x, w, bias=b.to(cuda), residual=r32.to(cuda), out_fp32=True, alpha=0.5)
    assert y32.dtype == torch.float32 and rel(y32, 0.5 * (xv @ wv.t()) + b.double() + r32.double()) < TOL_F32


def test_gemm_two_sources_group_bias_geglu_and_gelu(cuda):
    from mudg_amd import ops
    from test_kernels_gpu import pack_geglu
    M, N, K1, K2 = 6 * 40, 96, 64, 128
    x1, v1 = operand(f32(M, K1, seed=1), cuda)
    x2, v2 = operand(f32(M, K2, seed=2), cuda)
    w, wv = operand(f32(N, K1 + K2, seed=3, scale=0.05), cuda)
    gb = f32(6, N, seed=4)
    ref = torch.cat([v1, v2], 1) @ wv.t() + gb.double().repeat_interleave(40, 0)
    y = ops.gemm(x1, w, x2=x2, gbias=gb.to(cuda), rows_per_group=40)
    assert rel(value(y), ref) < TOL_OP
    # GEGLU (value * gelu_erf(gate)) and the plain GELU epilogue
    C = 64
    x, xv = operand(f32(300, C, seed=5), cuda)
    wf, bf = f32(8 * C, C, seed=6, scale=0.1), f32(8 * C, seed=7, scale=0.1)
    wp, bp = pack_geglu(wf, bf)
    wg, wgv = operand(wp, cuda)
    h = xv @ wgv.t() + bp.double()                                   # packed order: blocks of [32 value | 32 gate]
    h = h.reshape(300, -1, 2, 32)
    ref = (h[:, :, 0] * F.gelu(h[:, :, 1])).reshape(300, -1)
    y = ops.gemm(x, wg, bias=bp.to(cuda), geglu=True)
    # the 16-bit builds evaluate the gate through a 1025-entry table (7e-6 absolute): inside their operand rounding
    assert tuple(y.shape) == (300, 4 * C) and rel(value(y), ref) < TOL_OP
    w2, w2v = operand(f32(96, C, seed=8, scale=0.1), cuda)
    y = ops.gemm(x, w2, gelu=True, out_fp32=True)
    assert rel(y, F.gelu(xv @ w2v.t())) < max(TOL_F32, 2e-6)


def test_gemm_swapped_batched_writes_v_transposed(cuda):
    from mudg_amd import ops
    frames, hw, C = 3, 77, 128
    x, xv = operand(f32(frames * hw, C, seed=1), cuda)
    wv_t, wv = operand(f32(C, C, seed=2, scale=0.1), cuda)
    ld = (hw + 7) // 8 * 8
    out = ops.empty_rows(frames * C, ld, None, cuda)
    ops.gemm(wv_t, x, out=out, batch=frames, sx=0, sw=hw * x.stride(0), sy=C * out.stride(0), M=C, N=hw, K=C)
    want = (xv @ wv.t()).reshape(frames, hw, C).transpose(1, 2)
    got = value(out).reshape(frames, C, ld)[:, :, :hw]
    assert rel(got, want) < TOL_OP


def _rows(x):            # (F, C, H, W) -> [F*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def _unrows(v, f, h, w):
    return v.reshape(f, h, w, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("cin,cout,h,w,stride,ups,slab", [(64, 96, 9, 16, 1, False, False), (64, 64, 10, 16, 2, False, True),
                                                          (72, 40, 5, 7, 2, False, False), (64, 128, 5, 8, 1, True, False),
                                                          (16, 320, 9, 16, 1, False, False), (128, 96, 10, 12, 1, False, True)])
def test_conv3x3(cuda, cin, cout, h, w, stride, ups, slab):
    from mudg_amd import ops
    from test_kernels_gpu import pack_conv, pack_conv_slab
    frames = 3
    x, xv = operand(_rows(f32(frames, cin, h, w, seed=1)), cuda)
    wt = f32(cout, cin, 3, 3, seed=2, scale=0.05)
    wp, wpv = operand(pack_conv_slab(wt) if slab else pack_conv(wt), cuda)
    b = f32(cout, seed=3)
    # rebuild the (Cout, Cin, 3, 3) filter from the values the packed operand actually holds
    if slab:
        wq = wpv.reshape(cout, cin // 64, 9, 64).permute(0, 2, 1, 3).reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
    else:
        wq = wpv.reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
    xin = _unrows(xv, frames, h, w)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, wq, b.double(), stride=stride, padding=1)
    y = ops.conv3x3(x, wp, frames=frames, hin=h, win=w, cin=cin, stride=stride, upsample=ups, bias=b.to(cuda),
                    korder=int(slab), stats=not ups)
    assert rel(_unrows(value(y), frames, ref.shape[2], ref.shape[3]), ref) < TOL_OP
    y32 = ops.conv3x3(x, wp, frames=frames, hin=h, win=w, cin=cin, stride=stride, upsample=ups, bias=b.to(cuda),
                      korder=int(slab), out_fp32=True)
    assert rel(_unrows(y32.double().cpu(), frames, ref.shape[2], ref.shape[3]), ref) < TOL_F32


def test_tconv3(cuda):
    from mudg_amd import ops
    B, T, H, W, cin, cout = 2, 5, 6, 8, 64, 96
    x5 = f32(B, cin, T, H, W, seed=1)
    x, xv = operand(x5.permute(0, 2, 3, 4, 1).reshape(-1, cin).contiguous(), cuda)
    wt = f32(cout, cin, 3, 1, 1, seed=2, scale=0.05)
    wp, wpv = operand(wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(cout, -1).contiguous(), cuda)
    b = f32(cout, seed=3)
    wq = wpv.reshape(cout, 3, cin).permute(0, 2, 1)[..., None, None]
    xin = xv.reshape(B, T, H, W, cin).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(xin, wq, b.double(), padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(-1, cout)
    y = ops.tconv3(x, wp, clips=B, t=T, hw=H * W, cin=cin, bias=b.to(cuda), out_fp32=True)
    assert rel(y, ref) < TOL_F32


def _attention_ref(qv, kv, vv, frames, heads, nq, nk, scale, kv_div=1):
    q = qv.reshape(frames, nq, heads, 64).permute(0, 2, 1, 3)
    k = kv.reshape(frames // kv_div, nk, heads, 64).permute(0, 2, 1, 3).repeat_interleave(kv_div, 0)
    v = vv.reshape(frames // kv_div, nk, heads, 64).permute(0, 2, 1, 3).repeat_interleave(kv_div, 0)
    p = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(frames * nq, heads * 64)


def _vt(vvals, batches, nk, c, cuda):
    """V^T operand [batches * C, ld] from V values [batches * nk, C]."""
    from mudg_amd import ops
    ld = (nk + 7) // 8 * 8
    out = ops.empty_rows(batches * c, ld, None, cuda)
    src = torch.zeros(batches * c, ld)
    src.reshape(batches, c, ld)[:, :, :nk] = vvals.float().reshape(batches, nk, c).transpose(1, 2)
    ops.cast_rows(src.to(cuda), out)
    return out


@pytest.mark.parametrize("frames,heads,nq,nk,kv_div", [(2, 2, 200, 200, 1), (3, 1, 640, 640, 1), (4, 2, 96, 77, 2),
                                                       (2, 3, 130, 16, 1)])
def test_attention(cuda, frames, heads, nq, nk, kv_div):
    from mudg_amd import ops
    c = heads * 64
    q, qv = operand(f32(frames * nq, c, seed=1), cuda)
    k, kv = operand(f32(frames // kv_div * nk, c, seed=2), cuda)
    vsrc, vv = operand(f32(frames // kv_div * nk, c, seed=3), cuda)
    vt = _vt(vv, frames // kv_div, nk, c, cuda)
    out = ops.empty_rows(frames * nq, c, None, cuda)
    ops.attention(q, k, vt, out, frames=frames, heads=heads, nq=nq, nk=nk, kv_div=kv_div, scale=0.125)
    ref = _attention_ref(qv, kv, vv, frames, heads, nq, nk, 0.125, kv_div)
    # the probabilities enter the second MFMA as operands: one more operand rounding than a GEMM output
    assert rel(value(out), ref) < 2 * TOL_OP
    first = value(out)
    ops.attention(q, k, vt, out, frames=frames, heads=heads, nq=nq, nk=nk, kv_div=kv_div, scale=0.125, accumulate=True)
    assert rel(value(out), 2 * first) < 2 * TOL_OP


@pytest.mark.parametrize("T", [16, 7, 32])
def test_temporal_attention(cuda, T):
    from mudg_amd import ops
    B, HW, heads = 2, 37, 3
    c = heads * 64
    qkv, v = operand(f32(B * T * HW, 3 * c, seed=1), cuda)
    out = ops.empty_rows(B * T * HW, c, None, cuda)
    ops.temporal_attention(qkv, out, clips=B, t=T, hw=HW, heads=heads, scale=0.125)
    x = v.reshape(B, T, HW, 3, heads, 64).permute(3, 0, 2, 4, 1, 5)           # (qkv, b, p, h, t, d)
    p = torch.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, dim=-1)
    ref = (p @ x[2]).permute(0, 3, 1, 2, 4).reshape(B * T * HW, c)
    assert rel(value(out), ref) < 2 * TOL_OP


@pytest.mark.parametrize("x_fp32", [True, False])
def test_groupnorm_and_layernorm(cuda, x_fp32):
    from mudg_amd import ops
    samples, rows, C = 3, 200, 96
    xs = f32(samples * rows, C, seed=1) * 2 + 0.5
    if x_fp32:
        x, xv = xs.to(cuda), xs.double()
    else:
        x, xv = operand(xs, cuda)
    g, b = 1 + 0.1 * f32(C, seed=2), 0.1 * f32(C, seed=3)
    for silu in (False, True):
        y = ops.groupnorm(x, g.to(cuda), b.to(cuda), samples=samples, rows=rows, eps=1e-5, silu=silu)
        ref = F.group_norm(xv.reshape(samples, rows, C).transpose(1, 2), 32, g.double(), b.double(), 1e-5).transpose(1, 2)
        ref = ref.reshape(-1, C)
        if silu:
            ref = ref * torch.sigmoid(ref)
        assert rel(value(y), ref) < TOL_OP
    y = ops.layernorm(x, g.to(cuda), b.to(cuda), eps=1e-5)
    assert rel(value(y), F.layer_norm(xv, (C,), g.double(), b.double(), 1e-5)) < TOL_OP


@pytest.mark.parametrize("cin,cout,frames,h,w", [(64, 96, 3, 5, 7), (128, 64, 2, 16, 16), (192, 320, 1, 9, 16), (64, 128, 5, 8, 4)])
@pytest.mark.parametrize("stream", [False, True])
@pytest.mark.skipif(os.environ.get("MUDG_GEMM_FAST") == "0", reason="the sub-pixel form runs on the descriptor loader only")
def test_upsample_conv_in_sub_pixel_form(cuda, cin, cout, frames, h, w, stream):
    """Nearest-2x upsample + 3x3 conv as four 2x2 convs on the low-resolution rows (MudgGemmDesc.subpixel): against
    conv2d(interpolate(x)) in fp64 on the values the operands hold, and against the upsampling loader fed the 3x3 weights.
    Ragged sizes: row counts that are no multiple of the 128-row tile or of the image width, Cout below / across the
    128-column tile, several frames per tile."""
    import torch.nn as nn
    from mudg_amd import ops
    from mudg_amd.engine import packing as pk, unet as U
    conv = nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(f32(cout, cin, 3, 3, seed=2, scale=0.05))
        conv.bias.copy_(f32(cout, seed=3))
    conv = conv.to(cuda)
    x, xv = operand(_rows(f32(frames, cin, h, w, seed=1)), cuda)
    wsub = pk.conv3x3_subpixel(conv)
    assert wsub is not None and tuple(wsub.shape) == (4 * cout, 4 * cin)
    y = ops.conv3x3_up2(x, wsub, frames=frames, hin=h, win=w, cin=cin, bias=pk.f32(conv, "bias"), out_stream=stream)
    assert y is not None and tuple(y.shape) == (frames * 4 * h * w, cout)
    assert y.dtype == (ops.STREAM() if stream else _hip.operand_dtype())
    up = F.interpolate(_unrows(xv, frames, h, w), scale_factor=2, mode="nearest")
    ref = _rows(F.conv2d(up, conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), padding=1))
    # the four summed weights are rounded to the operand precision AFTER the sum: one more operand rounding than TOL_OP prices
    assert rel(value(y) if not stream else y, ref) < 2 * TOL_OP + (4e-4 if stream and y.dtype == torch.float16 else 0)
    w3, cpad, korder = pk.conv3x3(conv)
    old = ops.conv3x3(x, w3, frames=frames, hin=h, win=w, cin=cpad, upsample=True, bias=pk.f32(conv, "bias"), korder=korder,
                      out_stream=stream)
    assert rel(value(y) if not stream else y, value(old) if not stream else old) < 3 * TOL_OP + (8e-4 if stream and y.dtype == torch.float16 else 0)
    assert torch.equal(U.upsample_conv(conv, x, frames, h, w, stream=stream), y) or not U._SUBPIXEL
    assert pk.conv3x3_subpixel(nn.Conv2d(72, 8, 3, padding=1).to(cuda)) is None           # Cin % 64 != 0: the caller falls back


@pytest.mark.parametrize("C", [320, 512, 640, 1024, 1280, 384])
def test_layernorm_at_the_unet_widths(cuda, C):
    """The UNet's LayerNorm widths run on the lanes-per-row kernel (8 or 16 lanes per row, several rows per wave); 384 stays
    on the one-wave-per-row kernel.  Ragged row counts (a partly filled last wave and workgroup), every input storage
    (operand, fp32, the residual stream's), rows independent of their neighbours."""
    from mudg_amd import ops
    g, b = 1 + 0.1 * f32(C, seed=2), 0.1 * f32(C, seed=3)
    for rows in (1, 37, 259):
        xs = f32(rows, C, seed=rows) * 1.5 + 0.25
        want = lambda xv: F.layer_norm(xv, (C,), g.double(), b.double(), 1e-5)
        xo, xov = operand(xs, cuda)
        y = ops.layernorm(xo, g.to(cuda), b.to(cuda), eps=1e-5)
        assert rel(value(y), want(xov)) < TOL_OP
        assert rel(value(ops.layernorm(xs.to(cuda), g.to(cuda), b.to(cuda), eps=1e-5)), want(xs.double())) < TOL_OP
        xh = xs.to(cuda).to(ops.STREAM())
        assert rel(value(ops.layernorm(xh, g.to(cuda), b.to(cuda), eps=1e-5)), want(xh.double().cpu())) < TOL_OP
        if rows > 1:      # a row's result does not depend on which rows share its wave
            head = ops.layernorm(xo[:rows - 1], g.to(cuda), b.to(cuda), eps=1e-5)
            assert torch.equal(value(head), value(y)[:rows - 1])


def test_groupnorm_fused_statistics_from_the_producing_conv(cuda):
    """A conv writes the next GroupNorm's partial sums (over the values it stored); the fused norm must agree with the
    two-pass one on the same tensor."""
    from mudg_amd import ops
    from test_kernels_gpu import pack_conv_slab
    frames, h, w, cin, cout = 2, 16, 16, 64, 64                      # 256 rows per frame: two 128-row blocks
    x, _ = operand(_rows(f32(frames, cin, h, w, seed=1)), cuda)
    wp, _ = operand(pack_conv_slab(f32(cout, cin, 3, 3, seed=2, scale=0.05)), cuda)
    g, b = 1 + 0.1 * f32(cout, seed=3), 0.1 * f32(cout, seed=4)
    for out_fp32 in (False, True):
        y = ops.conv3x3(x, wp, frames=frames, hin=h, win=w, cin=cin, korder=1, stats=True, out_fp32=out_fp32)
        fused = ops.groupnorm(y, g.to(cuda), b.to(cuda), samples=frames, rows=h * w, eps=1e-5, silu=True)
        plain = ops.groupnorm(y, g.to(cuda), b.to(cuda), samples=frames, rows=h * w, eps=1e-5, silu=True, fused=False)
        assert rel(value(fused), value(plain)) < max(TOL_OP, 1e-6)


# the 288 x 320 tile (csrc/wgemm.hip): the 16-bit builds and bf16x3 (a variant child may switch it off: tests/test_gemm_variants_gpu.py)
WIDE_BUILD = _hip.planes() <= 2 and not (os.environ.get("MUDG_DEBUG_VARIANTS") == "1" and os.environ.get("MUDG_GEMM_W288") == "0")


@pytest.mark.parametrize("res_kind,korder", [("operand", 1), ("f32", 0), ("none", 1)])
def test_wide288_conv_in_every_operand_mode(cuda, res_kind, korder):
    """Frames of 576 = 2 x 288 pixels: the library's rule sends the 3x3 conv to the 288 x 320 tile (bf16x3: both pieces of both operands
    per k half, three MFMAs per fragment pair).  Two channel sources, bias + per-frame group bias, a residual, GroupNorm partials per
    288-row block of what was stored, and a frame's rows do not depend on the batch."""
    from mudg_amd import ops
    from test_kernels_gpu import pack_conv, pack_conv_slab
    f, h, wd, c1, c2, cout = 3, 24, 24, 64, 128, 320
    cin, M = c1 + c2, 3 * 24 * 24
    xa, xav = operand(_rows(f32(f, c1, h, wd, seed=1)), cuda)
    xb, xbv = operand(_rows(f32(f, c2, h, wd, seed=2)), cuda)
    wt = f32(cout, cin, 3, 3, seed=3, scale=0.03)
    wp, wpv = operand(pack_conv_slab(wt) if korder else pack_conv(wt), cuda)
    wq = (wpv.reshape(cout, cin // 64, 9, 64).permute(0, 2, 1, 3).reshape(cout, 3, 3, cin) if korder else wpv.reshape(cout, 3, 3, cin)).permute(0, 3, 1, 2)
    b, gb = f32(cout, seed=4, scale=0.1), f32(f, cout, seed=5)
    r32 = f32(M, cout, seed=6)
    res, resv = (None, 0.0) if res_kind == "none" else (operand(r32, cuda) if res_kind == "operand" else (r32.to(cuda), r32.double()))
    xin = _unrows(torch.cat([xav, xbv], 1), f, h, wd)
    ref = _rows(F.conv2d(xin, wq, b.double(), padding=1)) + gb.double().repeat_interleave(h * wd, 0) + resv
    kw = dict(frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b.to(cuda), rows_per_group=h * wd, stats=True, out_fp32=res_kind == "f32")
    y = ops.conv3x3(xa, wp, x2=xb, gbias=gb.to(cuda), residual=res, **kw)
    assert rel(value(y), ref) < (TOL_F32 if res_kind == "f32" else TOL_OP)
    rows = getattr(y, ops.GN_ATTR + "_rows")
    assert rows == (288 if WIDE_BUILD else 128)
    stats = getattr(y, ops.GN_ATTR)
    if M % rows == 0:
        yv = value(y).reshape(-1, rows, cout)
        assert rel(stats[..., 0], yv.sum(1)) < 1e-5 and rel(stats[..., 1], (yv * yv).sum(1)) < 1e-5
    n1 = h * wd
    one = ops.conv3x3(xa[:n1], wp, x2=xb[:n1], gbias=gb[:1].to(cuda), residual=None if res is None else res[:n1], **dict(kw, frames=1))
    assert torch.equal(value(one), value(y)[:n1]) and (n1 % rows or torch.equal(getattr(one, ops.GN_ATTR), stats[:n1 // rows]))


def test_wide288_gemm_and_temporal_conv_in_every_operand_mode(cuda):
    """A plain GEMM with a frame hint of whole 288-row tiles, ragged in M (K = 2560: on the 288 x 320 tile under every build's rule), and a
    temporal conv in the plain K order on clips of 4 frames x 288 pixels (first / last frame of a clip see zero padding)."""
    from mudg_amd import ops
    M, N, K = 288 * 5 + 100, 320, 2560
    x, xv = operand(f32(M, K, seed=1), cuda)
    w, wv = operand(f32(N, K, seed=2, scale=0.02), cuda)
    b, r32 = f32(N, seed=3, scale=0.1), f32(M, N, seed=4)
    y = ops.gemm(x, w, bias=b.to(cuda), residual=r32.to(cuda), stats=True, frame_rows=288)
    assert rel(value(y), xv @ wv.t() + b.double() + r32.double()) < TOL_OP
    rows = getattr(y, ops.GN_ATTR + "_rows")
    assert rows == (288 if WIDE_BUILD else 128) and getattr(y, ops.GN_ATTR).shape[0] == (M + rows - 1) // rows
    y32 = ops.gemm(x, w, bias=b.to(cuda), out_fp32=True, frame_rows=288)
    assert rel(y32, xv @ wv.t() + b.double()) < TOL_F32
    assert torch.equal(ops.gemm(x[:288 * 2], w, bias=b.to(cuda), out_fp32=True, frame_rows=288), y32[:288 * 2])
    # GEGLU with a frame hint of whole tiles (16-bit builds: the table lookup; bf16x3: the cubic through the same table)
    from test_kernels_gpu import pack_geglu
    C = 128
    xg, xgv = operand(f32(288 * 3, C, seed=11), cuda)
    wf, bf_ = f32(8 * C, C, seed=12, scale=0.1), f32(8 * C, seed=13, scale=0.1)
    wpk, bpk = pack_geglu(wf, bf_)
    wg, wgv = operand(wpk, cuda)
    hh = (xgv @ wgv.t() + bpk.double()).reshape(288 * 3, -1, 2, 32)
    yg = ops.gemm(xg, wg, bias=bpk.to(cuda), geglu=True, frame_rows=288)
    assert rel(value(yg), (hh[:, :, 0] * F.gelu(hh[:, :, 1])).reshape(288 * 3, -1)) < TOL_OP
    clips, t, hw, c, co = 2, 4, 288, 128, 640
    xt, xtv = operand(f32(clips * t * hw, c, seed=5), cuda)
    wt, wtv = operand(f32(co, 3 * c, seed=6, scale=0.05), cuda)
    bt = f32(co, seed=7, scale=0.1)
    xi = xtv.reshape(clips, t, hw, c).permute(0, 3, 1, 2)                                       # (clip, c, t, hw)
    wk = wtv.reshape(co, 3, c).permute(0, 2, 1)                                                # (co, c, tap)
    ref = F.conv2d(xi, wk[..., None], padding=(1, 0)).permute(0, 2, 3, 1).reshape(clips * t * hw, co) + bt.double()
    assert ops.tconv3_wide(t, hw, c, co) == WIDE_BUILD
    yt = ops.tconv3(xt, wt, clips=clips, t=t, hw=hw, cin=c, bias=bt.to(cuda), stats=True)
    assert rel(value(yt), ref) < TOL_OP
    rows = getattr(yt, ops.GN_ATTR + "_rows")
    yv = value(yt).reshape(-1, rows, co)
    assert rel(getattr(yt, ops.GN_ATTR)[..., 0], yv.sum(1)) < 1e-5
    one = ops.tconv3(xt[:t * hw], wt, clips=1, t=t, hw=hw, cin=c, bias=bt.to(cuda), stats=True)
    assert torch.equal(value(one), value(yt)[:t * hw])


def test_softmax_rows_and_layout_kernels(cuda):
    from mudg_amd import ops
    s = f32(50, 333, seed=1) * 3
    p = ops.softmax_rows(s.to(cuda))
    assert rel(value(p), torch.softmax(s.double(), -1)) < TOL_OP
    x = f32(2, 5, 3, 4, 6, seed=2)                                        # (b c t h w)
    rows = ops.empty_rows(2 * 3 * 24, 8, None, cuda)
    ops.ncthw_to_rows(x.to(cuda), rows, 0)
    ops.zero_channels(rows, 5, 8)
    v = value(rows)
    want = x.permute(0, 2, 3, 4, 1).reshape(-1, 5).double()
    assert rel(v[:, :5], want) < TOL_OP and float(v[:, 5:].abs().max()) == 0.0
    back = ops.rows_to_ncthw(rows, (2, 5, 3, 4, 6))
    assert rel(back, x) < TOL_OP
    dst = ops.empty_rows(2 * 3 * 24, 8, None, cuda)
    ops.copy_rows(rows, dst)
    assert torch.equal(value(dst), v)


def test_residual_stream_storage(cuda):
    """The residual stream's storage (ops.STREAM(): fp16 in the 16-bit builds, fp32 in the split builds) through the
    kernels that touch it: GEMM epilogue reading a stream residual and writing a stream result (+ GroupNorm partials of
    what it stored), GroupNorm / LayerNorm reading it, the stream -> operand cast and the boundary layout kernel."""
    from mudg_amd import ops
    S = ops.STREAM()
    assert S == (torch.float16 if MODE == "bf16" else torch.float32)
    tol_s = 1e-6 if S == torch.float32 else 4e-4          # one fp16 rounding of the result
    M, N, K = 512, 96, 128
    x, xv = operand(f32(M, K, seed=1), cuda)
    w, wv = operand(f32(N, K, seed=2, scale=0.1), cuda)
    r = (f32(M, N, seed=3) * 3).to(cuda).to(S)
    y = ops.gemm(x, w, residual=r, out_stream=True, stats=True)
    assert y.dtype == S
    ref = xv @ wv.t() + r.double().cpu()
    assert rel(y, ref) < max(tol_s, TOL_F32)
    g, b = 1 + 0.1 * f32(N, seed=4), 0.1 * f32(N, seed=5)
    yv = y.double().cpu()
    want = F.group_norm(yv.reshape(2, 256, N).transpose(1, 2), 32, g.double(), b.double(), 1e-5).transpose(1, 2).reshape(-1, N)
    fused = ops.groupnorm(y, g.to(cuda), b.to(cuda), samples=2, rows=256, eps=1e-5, silu=False)
    plain = ops.groupnorm(y, g.to(cuda), b.to(cuda), samples=2, rows=256, eps=1e-5, silu=False, fused=False)
    assert rel(value(fused), want) < TOL_OP and rel(value(plain), want) < TOL_OP
    assert rel(value(ops.layernorm(y, g.to(cuda), b.to(cuda))), F.layer_norm(yv, (N,), g.double(), b.double(), 1e-5)) < TOL_OP
    op = ops.cast_bf16(y)                                  # stream -> MFMA operand (skip / downsample conv inputs)
    assert op.dtype == _hip.operand_dtype() and rel(value(op), yv) < TOL_OP
    rows = torch.randn(2 * 3 * 20, 8, device=cuda).to(S)
    back = ops.rows_to_ncthw(rows, (2, 8, 3, 4, 5))
    assert torch.equal(back.cpu(), rows.float().cpu().reshape(2, 3, 20, 8).permute(0, 3, 1, 2).reshape(2, 8, 3, 4, 5))


def test_attention_two_key_sets_in_one_launch(cuda):
    """softmax(q k_t^T) v_t + softmax(q k_i^T) v_i (text + image cross-attention, attention.py:128-142) in ONE launch
    against the fp64 reference and against the two-launch (accumulate) form."""
    from mudg_amd import ops
    frames, heads, nq, T = 4, 2, 150, 2
    c = heads * 64
    q, qv = operand(f32(frames * nq, c, seed=1), cuda)
    kt, ktv = operand(f32(frames // T * 77, c, seed=2), cuda)
    _, vtv = operand(f32(frames // T * 77, c, seed=3), cuda)
    ki, kiv = operand(f32(frames * 16, c, seed=4), cuda)
    _, viv = operand(f32(frames * 16, c, seed=5), cuda)
    vt_t, vt_i = _vt(vtv, frames // T, 77, c, cuda), _vt(viv, frames, 16, c, cuda)
    one = ops.empty_rows(frames * nq, c, None, cuda)
    ops.attention(q, kt, vt_t, one, frames=frames, heads=heads, nq=nq, nk=77, kv_div=T, scale=0.125, k2=ki, vt2=vt_i, nk2=16)
    ref = _attention_ref(qv, ktv, vtv, frames, heads, nq, 77, 0.125, T) + _attention_ref(qv, kiv, viv, frames, heads, nq, 16, 0.125, 1)
    assert rel(value(one), ref) < 2 * TOL_OP
    two = ops.empty_rows(frames * nq, c, None, cuda)
    ops.attention(q, kt, vt_t, two, frames=frames, heads=heads, nq=nq, nk=77, kv_div=T, scale=0.125)
    ops.attention(q, ki, vt_i, two, frames=frames, heads=heads, nq=nq, nk=16, scale=0.125, accumulate=True)
    assert rel(value(one), value(two)) < 2 * TOL_OP


def test_cross_attention_with_resident_key_tiles_walks_many_query_tiles(cuda):
    """The cross-attention of the fine levels: 77 text + 16 image keys, thousands of 128-query tiles.  In the 16-bit builds a workgroup
    stages the three key tiles of a (frame, head) once and walks several query tiles (xattn_kernel); a sub-batch small enough for the
    one-tile-per-workgroup kernel gives the same bits, and both match the fp64 reference.  One set and two sets; a ragged last tile."""
    from mudg_amd import ops
    frames, heads, nq, T = 8, 5, 128 * 52 + 40, 4                      # 53 x 40 = 2120 query tiles: two per workgroup
    c = heads * 64
    q, qv = operand(f32(frames * nq, c, seed=1), cuda)
    kt, ktv = operand(f32(frames // T * 77, c, seed=2), cuda)
    _, vtv = operand(f32(frames // T * 77, c, seed=3), cuda)
    ki, kiv = operand(f32(frames * 16, c, seed=4), cuda)
    _, viv = operand(f32(frames * 16, c, seed=5), cuda)
    vt_t, vt_i = _vt(vtv, frames // T, 77, c, cuda), _vt(viv, frames, 16, c, cuda)
    both = ops.empty_rows(frames * nq, c, None, cuda)
    ops.attention(q, kt, vt_t, both, frames=frames, heads=heads, nq=nq, nk=77, kv_div=T, scale=0.125, k2=ki, vt2=vt_i, nk2=16)
    sel = torch.arange(0, frames * nq, 97)
    ref = _attention_ref(qv, ktv, vtv, frames, heads, nq, 77, 0.125, T) + _attention_ref(qv, kiv, viv, frames, heads, nq, 16, 0.125, 1)
    assert rel(value(both)[sel], ref[sel]) < 2 * TOL_OP
    one = ops.empty_rows(frames * nq, c, None, cuda)
    ops.attention(q, kt, vt_t, one, frames=frames, heads=heads, nq=nq, nk=77, kv_div=T, scale=0.125)
    assert rel(value(one)[sel], _attention_ref(qv, ktv, vtv, frames, heads, nq, 77, 0.125, T)[sel]) < TOL_OP
    # frame 0 alone: 53 x 5 query tiles, the one-tile kernel — the same bits (it shares its text keys with nobody else now: kv_div 1)
    sub2, sub1 = ops.empty_rows(nq, c, None, cuda), ops.empty_rows(nq, c, None, cuda)
    ops.attention(q[:nq], kt[:77], vt_t[:c], sub2, frames=1, heads=heads, nq=nq, nk=77, scale=0.125, k2=ki[:16], vt2=vt_i[:c], nk2=16)
    ops.attention(q[:nq], kt[:77], vt_t[:c], sub1, frames=1, heads=heads, nq=nq, nk=77, scale=0.125)
    assert torch.equal(value(sub2), value(both)[:nq]) and torch.equal(value(sub1), value(one)[:nq])


@pytest.mark.skipif(_hip.planes() > 1, reason="the lean softmax belongs to the 16-bit builds' long self-attention kernel")
@pytest.mark.parametrize("spread", [1.0, 4.0, 12.0])
def test_attention_lean_softmax_with_prescaled_q(cuda, spread):
    """q_prescaled: Q carries scale * log2(e), the kernel exponentiates Q K^T directly with the first tile's row maximum
    as the accumulators' initial value.  spread = 12 puts later scores far above the first tile's maximum (the deferred
    reference has to cope; beyond 2^40 the classic loop takes over — forced here by a third, extreme case); spread = 4 puts
    them 2^16 ... 2^30 above it: inside the bf16 kernel's lean range, beyond what an IEEE-half P can hold — the fp16 build
    must take the classic loop there (its limit is 2^15) instead of storing inf."""
    import math
    from mudg_amd import ops
    frames, heads, n = 2, 2, 768
    c = heads * 64
    cl2 = 0.125 * math.log2(math.e)
    qs = f32(frames * n, c, seed=1)
    ks = f32(frames * n, c, seed=2) * spread
    ks[n // 2:] *= 1.5                                    # later keys score higher than the first tile's
    q, qv = operand(qs * cl2, cuda)
    k, kv = operand(ks, cuda)
    _, vv = operand(f32(frames * n, c, seed=3), cuda)
    vt = _vt(vv, frames, n, c, cuda)
    out = ops.empty_rows(frames * n, c, None, cuda)
    ops.attention(q, k, vt, out, frames=frames, heads=heads, nq=n, nk=n, q_prescaled=True)
    ref = _attention_ref(qv, kv, vv, frames, heads, n, n, math.log(2.0))       # softmax of 2^(q k)
    assert rel(value(out), ref) < 2 * TOL_OP
    assert torch.isfinite(value(out)).all()
    if spread > 4:      # extreme: one key per row 2^60 above everything in the first tile -> the classic loop redoes the block
        kbig = ks.clone()
        kbig[n - 1] = qs[0] * 50.0
        k2, k2v = operand(kbig, cuda)
        ops.attention(q, k2, vt, out, frames=frames, heads=heads, nq=n, nk=n, q_prescaled=True)
        assert torch.isfinite(value(out)).all()
        assert rel(value(out), _attention_ref(qv, k2v, vv, frames, heads, n, n, math.log(2.0))) < 2 * TOL_OP


@pytest.mark.skipif(_hip.planes() > 1, reason="fp8 scores belong to the 16-bit builds")
@pytest.mark.skipif(__import__("os").environ.get("MUDG_ATTN_Q") == "32" or __import__("os").environ.get("MUDG_ATTN_DMA") == "0",
                    reason="the fp8 score path lives in the LDS-DMA staged 64-query kernel, which this variant switch turns off")
def test_mxfp8_quantiser_and_fp8_score_attention(cuda):
    """BASELINE config 5's kernel: OCP MX-fp8 quantisation (e4m3 + one E8M0 scale per 32 dims) checked against a numpy-style
    restatement, and the long self-attention with Q K^T on the fp8 MFMA against the fp64 softmax of the DEQUANTISED q / k
    (so the kernel's arithmetic is what is measured; the quantisation error itself is printed against the bf16 inputs)."""
    import math
    from mudg_amd import ops
    frames, heads, n = 2, 2, 640
    c = heads * 64
    cl2 = 0.125 * math.log2(math.e)
    qk_src = torch.cat([f32(frames * n, c, seed=1) * cl2, f32(frames * n, c, seed=2)], 1)
    qk, qkv = operand(qk_src, cuda)
    y8, s8 = ops.quantize_mxfp8(qk)
    # restatement: block amax -> shared exponent floor(log2 amax) - 8 -> e4m3 round-to-nearest-even of x / 2^E, saturating
    blocks = qkv.reshape(frames * n, -1, 32)
    amax = blocks.abs().amax(-1)
    E = torch.where(amax > 0, torch.floor(torch.log2(amax.clamp_min(1e-300))) - 8, torch.zeros_like(amax))
    assert torch.equal(s8.cpu().to(torch.int64), (E + 127).to(torch.int64).reshape(frames * n, -1))
    want = (blocks / torch.pow(2.0, E)[..., None]).clamp(-448, 448).float().to(torch.float8_e4m3fn).float().reshape(frames * n, -1)
    got = y8.cpu().view(torch.float8_e4m3fn).float()
    assert torch.equal(got, want)
    # the same copy written by a GEMM's own epilogue (MudgGemmDesc.Y8: how the engine makes it) equals the quantiser of its result
    xg, _ = operand(f32(1000, 64, seed=7), cuda)
    wg, _ = operand(f32(2 * c, 64, seed=8) * 0.2, cuda)
    yg, yg8, sg8 = ops.gemm(xg, wg, fp8=True)
    ref8, refs = ops.quantize_mxfp8(yg)
    assert torch.equal(yg8, ref8) and torch.equal(sg8, refs)
    deq = (got.reshape(frames * n, -1, 32).double() * torch.pow(2.0, E)[..., None]).reshape(frames * n, -1)
    print(f"MX-fp8 quantisation error of q|k: rel-L2 {rel(deq, qkv):.3e}")
    _, vv = operand(f32(frames * n, c, seed=3), cuda)
    vt = _vt(vv, frames, n, c, cuda)
    out = ops.empty_rows(frames * n, c, None, cuda)
    fp8 = (y8[:, :c], s8[:, :c // 32], y8[:, c:], s8[:, c // 32:])
    ops.attention(qk[:, :c], qk[:, c:], vt, out, frames=frames, heads=heads, nq=n, nk=n, q_prescaled=True, fp8=fp8)
    ref = _attention_ref(deq[:, :c], deq[:, c:], vv, frames, heads, n, n, math.log(2.0))
    assert rel(value(out), ref) < 2 * TOL_OP
    exact = _attention_ref(qkv[:, :c], qkv[:, c:], vv, frames, heads, n, n, math.log(2.0))
    print(f"fp8-score attention vs exact attention on the bf16 q / k: rel-L2 {rel(value(out), exact):.3e}")
