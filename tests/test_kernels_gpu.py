"""Per-kernel parity on the GPU: each C-ABI entry point against plain fp32 PyTorch ops on the CPU, fed the same
bf16-rounded inputs.  Tolerances: a bf16 result carries one rounding (relative 2^-9, rms ~1.1e-3), so bf16 outputs
are held to rel-L2 <= 3e-3; fp32 outputs to 2e-5 (accumulation order only)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mudg_amd import hip as _hip

BF = _hip.operand_dtype()       # bf16, or fp16 when the suite runs with MUDG_OPERAND=fp16
TOL_BF16 = 3e-3
TOL_F32 = 2e-5


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def pack_geglu(w, b):
    """Reference layout [value rows | gate rows] -> blocks of 32 value rows followed by their 32 gate rows."""
    n2 = w.shape[0] // 2
    idx = []
    for blk in range(n2 // 32):
        idx += list(range(blk * 32, blk * 32 + 32)) + list(range(n2 + blk * 32, n2 + blk * 32 + 32))
    idx = torch.tensor(idx)
    return w[idx].contiguous(), (b[idx].contiguous() if b is not None else None)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 96, 72), (1000, 320, 320), (77, 1280, 1024), (300, 4, 320),
                                   (129, 200, 8)])
def test_gemm_plain(cuda, M, N, K):
    from mudg_amd import ops
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3))
    r = rnd(M, N, seed=4)
    ref = x.float() @ w.float().t() + b + r.float()
    y = ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda), residual=r.to(cuda))
    assert y.dtype == BF and tuple(y.shape) == (M, N)
    assert rel_l2(y, ref) < TOL_BF16
    y32 = ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda), out_fp32=True, alpha=0.5)
    assert rel_l2(y32, 0.5 * (x.float() @ w.float().t()) + b) < TOL_F32


def test_gemm_transpose_detecting(cuda):
    """A = I with an asymmetric B catches a swapped row/column mapping in the MFMA epilogue."""
    from mudg_amd import ops
    n = 128
    x = torch.eye(n).to(BF)
    w = (torch.arange(n * n).reshape(n, n) % 251).float().to(BF)
    y = ops.gemm(x.to(cuda), w.to(cuda), out_fp32=True)
    assert torch.equal(y.cpu(), w.float().t())


def test_gemm_group_bias_and_two_sources(cuda):
    from mudg_amd import ops
    M, N, K1, K2 = 6 * 40, 96, 64, 128
    x1, x2, w = rnd(M, K1, seed=1), rnd(M, K2, seed=2), rnd(N, K1 + K2, seed=3, scale=0.05)
    gb = torch.randn(6, N, generator=torch.Generator().manual_seed(4))
    ref = torch.cat([x1, x2], 1).float() @ w.float().t() + gb.repeat_interleave(40, 0)
    y = ops.gemm(x1.to(cuda), w.to(cuda), x2=x2.to(cuda), gbias=gb.to(cuda), rows_per_group=40)
    assert rel_l2(y, ref) < TOL_BF16


def test_gemm_geglu(cuda):
    from mudg_amd import ops
    M, C = 300, 64
    x, w = rnd(M, C, seed=1), rnd(8 * C, C, seed=2, scale=0.1)
    b = torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1
    h = x.float() @ w.float().t() + b
    val, gate = h.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    wp, bp = pack_geglu(w, b)
    y = ops.gemm(x.to(cuda), wp.to(cuda), bias=bp.to(cuda), geglu=True)
    assert tuple(y.shape) == (M, 4 * C)
    assert rel_l2(y, ref) < TOL_BF16


def test_gemm_batched_swapped_gives_v_transposed(cuda):
    """The V projection is issued with operands swapped so that it writes V^T per frame (ld padded to 8)."""
    from mudg_amd import ops
    frames, hw, C = 3, 77, 128
    x, wv = rnd(frames * hw, C, seed=1), rnd(C, C, seed=2, scale=0.1)
    ld = (hw + 7) // 8 * 8
    out = torch.zeros(frames * C, ld, dtype=BF, device=cuda)
    ops.gemm(wv.to(cuda), x.to(cuda), out=out, batch=frames, sx=0, sw=hw * C, sy=C * ld, M=C, N=hw, K=C, ldy=ld)
    v = (x.float() @ wv.float().t()).reshape(frames, hw, C)
    got = out.reshape(frames, C, ld)[:, :, :hw]
    assert rel_l2(got, v.transpose(1, 2)) < TOL_BF16


def pack_conv(w):
    """(Cout, Cin, 3, 3) -> [Cout][tap][Cin]"""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def pack_conv_slab(w):
    """(Cout, Cin, 3, 3) with Cin % 64 == 0 -> [Cout][Cin/64][tap][64] (korder 1)"""
    co, ci = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(co, 9, ci // 64, 64).permute(0, 2, 1, 3).reshape(co, -1).contiguous()


def to_rows(x):       # (F, C, H, W) -> rows
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def from_rows(y, f, h, w):
    return y.reshape(f, h, w, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("cin,cout,h,w,stride,ups", [(64, 96, 9, 16, 1, False), (64, 64, 10, 16, 2, False),
                                                     (72, 40, 5, 7, 2, False), (64, 128, 5, 8, 1, True),
                                                     (16, 320, 9, 16, 1, False), (320, 4, 9, 16, 1, False)])
def test_conv3x3(cuda, cin, cout, h, w, stride, ups):
    from mudg_amd import ops
    frames = 3
    x = rnd(frames, cin, h, w, seed=1)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=0.05)
    b = torch.randn(cout, generator=torch.Generator().manual_seed(3))
    xin = F.interpolate(x.float(), scale_factor=2, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, wt.float(), b, stride=stride, padding=1)
    y = ops.conv3x3(to_rows(x).to(cuda), pack_conv(wt).to(cuda), frames=frames, hin=h, win=w, cin=cin, stride=stride,
                    upsample=ups, bias=b.to(cuda))
    got = from_rows(y.cpu().float(), frames, ref.shape[2], ref.shape[3])
    assert rel_l2(got, ref) < TOL_BF16


@pytest.mark.parametrize("cin,cout,stride,ups", [(128, 96, 1, False), (192, 64, 2, False), (64, 64, 1, True)])
def test_conv3x3_slab_major_k_order(cuda, cin, cout, stride, ups):
    from mudg_amd import ops
    frames, h, w = 2, 10, 12
    x = rnd(frames, cin, h, w, seed=1)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=0.05)
    xin = F.interpolate(x.float(), scale_factor=2, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, wt.float(), None, stride=stride, padding=1)
    y = ops.conv3x3(to_rows(x).to(cuda), pack_conv_slab(wt).to(cuda), frames=frames, hin=h, win=w, cin=cin,
                    stride=stride, upsample=ups, korder=1)
    assert rel_l2(from_rows(y.cpu().float(), frames, ref.shape[2], ref.shape[3]), ref) < TOL_BF16


def test_conv3x3_fused_epilogue_and_concat(cuda):
    from mudg_amd import ops
    frames, c1, c2, cout, h, w = 4, 64, 32, 64, 6, 8
    xa, xb = rnd(frames, c1, h, w, seed=1), rnd(frames, c2, h, w, seed=2)
    wt = rnd(cout, c1 + c2, 3, 3, seed=3, scale=0.05)
    b = torch.randn(cout, generator=torch.Generator().manual_seed(4))
    emb = torch.randn(2, cout, generator=torch.Generator().manual_seed(5))      # one row per clip of 2 frames
    res = rnd(frames, cout, h, w, seed=6)
    ref = F.conv2d(torch.cat([xa, xb], 1).float(), wt.float(), b, padding=1)
    ref = ref + emb.repeat_interleave(2, 0)[:, :, None, None] + res.float()
    y = ops.conv3x3(to_rows(xa).to(cuda), pack_conv(wt).to(cuda), frames=frames, hin=h, win=w, cin=c1 + c2,
                    x2=to_rows(xb).to(cuda), bias=b.to(cuda), gbias=emb.to(cuda), rows_per_group=2 * h * w,
                    residual=to_rows(res).to(cuda))
    assert rel_l2(from_rows(y.cpu().float(), frames, h, w), ref) < TOL_BF16


def test_tconv3(cuda):
    from mudg_amd import ops
    clips, t, h, w, c = 2, 5, 3, 4, 64
    x = rnd(clips, c, t, h, w, seed=1)
    wt = rnd(c, c, 3, 1, 1, seed=2, scale=0.05)
    b = torch.randn(c, generator=torch.Generator().manual_seed(3))
    ref = F.conv3d(x.float(), wt.float(), b, padding=(1, 0, 0)) + x.float()
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, c).contiguous()
    wp = wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, 3 * c).contiguous()
    y = ops.tconv3(rows.to(cuda), wp.to(cuda), clips=clips, t=t, hw=h * w, cin=c, bias=b.to(cuda),
                   residual=rows.to(cuda))
    got = y.cpu().float().reshape(clips, t, h, w, c).permute(0, 4, 1, 2, 3)
    assert rel_l2(got, ref) < TOL_BF16


def attn_ref(q, k, v, heads, scale):
    """q [F, Nq, C], k/v [Fk, Nk, C] fp32, frames share k/v in blocks."""
    f, nq, c = q.shape
    rep = f // k.shape[0]
    k, v = k.repeat_interleave(rep, 0), v.repeat_interleave(rep, 0)
    qh = q.reshape(f, nq, heads, 64).transpose(1, 2)
    kh = k.reshape(f, -1, heads, 64).transpose(1, 2)
    vh = v.reshape(f, -1, heads, 64).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(f, nq, c)


def make_vt(v, heads):
    fk, nk, c = v.shape
    ld = (nk + 7) // 8 * 8
    vt = torch.full((fk, c, ld), float("nan"), dtype=BF)     # padding must never be read as data
    vt[:, :, :nk] = v.transpose(1, 2)
    return vt, ld


@pytest.mark.parametrize("frames,heads,nq,nk,kv_div", [(2, 2, 200, 200, 1), (1, 5, 384, 384, 1), (4, 2, 150, 77, 2),
                                                       (3, 1, 70, 16, 1), (1, 1, 128, 64, 1),
                                                       (2, 3, 700, 333, 1), (2, 2, 1024, 1100, 2)])     # 64-queries-per-wave kernel, ragged
def test_attention(cuda, frames, heads, nq, nk, kv_div):
    from mudg_amd import ops
    c = heads * 64
    q = rnd(frames, nq, c, seed=1)
    k = rnd(frames // kv_div, nk, c, seed=2)
    v = rnd(frames // kv_div, nk, c, seed=3)
    ref = attn_ref(q.float(), k.float(), v.float(), heads, 0.125)
    vt, ld = make_vt(v, heads)
    out = torch.zeros(frames * nq, c, dtype=BF, device=cuda)
    ops.attention(q.reshape(-1, c).to(cuda), k.reshape(-1, c).to(cuda), vt.to(cuda), out, frames=frames, heads=heads,
                  nq=nq, nk=nk, ldvt=ld, svt=c * ld, kv_div=kv_div)
    assert rel_l2(out.cpu().reshape(frames, nq, c), ref) < TOL_BF16
    # second softmax accumulated on top (text + image cross-attention)
    ops.attention(q.reshape(-1, c).to(cuda), k.reshape(-1, c).to(cuda), vt.to(cuda), out, frames=frames, heads=heads,
                  nq=nq, nk=nk, ldvt=ld, svt=c * ld, kv_div=kv_div, accumulate=True)
    assert rel_l2(out.cpu().reshape(frames, nq, c), 2 * ref) < 2 * TOL_BF16


def test_attention_rescale_branch(cuda):
    """Spike one key per 64-key tile so that the running max jumps at every tile (online-softmax rescale path)."""
    from mudg_amd import ops
    heads, n = 1, 512
    q = rnd(1, n, 64, seed=1)
    k = rnd(1, n, 64, seed=2)
    v = rnd(1, n, 64, seed=3)
    for tile in range(n // 64):
        k[0, tile * 64 + 5] = (q[0, 7].float() * (1.0 + 0.5 * tile)).to(BF)
    ref = attn_ref(q.float(), k.float(), v.float(), heads, 0.125)
    vt, ld = make_vt(v, heads)
    out = torch.zeros(n, 64, dtype=BF, device=cuda)
    ops.attention(q.reshape(-1, 64).to(cuda), k.reshape(-1, 64).to(cuda), vt.to(cuda), out, frames=1, heads=1, nq=n,
                  nk=n, ldvt=ld, svt=64 * ld)
    assert rel_l2(out.cpu().reshape(1, n, 64), ref) < TOL_BF16
    assert float((out.cpu().float().reshape(1, n, 64) - ref).abs().max()) < 0.05


@pytest.mark.parametrize("clips,t,hw,heads", [(2, 16, 10, 3), (1, 5, 7, 2), (1, 20, 5, 1), (1, 16, 33, 8)])
def test_temporal_attention(cuda, clips, t, hw, heads):
    from mudg_amd import ops
    c = heads * 64
    qkv = rnd(clips * t * hw, 3 * c, seed=1)
    x = qkv.float().reshape(clips, t, hw, 3, heads, 64)
    q, k, v = (x[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))        # (b, hw, heads, t, 64)
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    ref = (p @ v).permute(0, 3, 1, 2, 4).reshape(clips * t * hw, c)
    out = torch.zeros(clips * t * hw, c, dtype=BF, device=cuda)
    ops.temporal_attention(qkv.to(cuda), out, clips=clips, t=t, hw=hw, heads=heads)
    assert rel_l2(out, ref) < TOL_BF16


@pytest.mark.parametrize("out_fp32", [False, True])
def test_conv_with_a_group_bias_does_not_depend_on_the_batch(cuda, out_fp32):
    """A ResBlock's conv adds bias + embedding term.  A 128-row tile inside one group adds them as one staged column constant, a tile
    that straddles groups adds them row by row — as the SAME pre-summed constant, so a row's bits do not depend on which tile it fell
    into (that changes with the number of clips in the launch: 36 frames of 4 x 4 pixels = 4.5 tiles against 12 frames = 1.5 tiles;
    seen as a one-ulp drift between stacked and back-to-back guidance passes before the two paths were aligned)."""
    from mudg_amd import ops
    cin, cout, h, w = 64, 128, 4, 4
    g = torch.Generator().manual_seed(21)
    for trial in range(4):
        x = rnd(36 * h * w, cin, seed=30 + trial).to(cuda)
        wt = rnd(cout, 9 * cin, seed=40 + trial, scale=0.05).to(cuda)
        bias, gb = torch.randn(cout, generator=g).to(cuda), torch.randn(9, cout, generator=g).to(cuda)
        ya = ops.conv3x3(x, wt, frames=36, hin=h, win=w, cin=cin, bias=bias, gbias=gb, rows_per_group=64, korder=1, out_fp32=out_fp32)
        yb = ops.conv3x3(x[:192].contiguous(), wt, frames=12, hin=h, win=w, cin=cin, bias=bias, gbias=gb[:3].contiguous(),
                         rows_per_group=64, korder=1, out_fp32=out_fp32)
        assert torch.equal(ya[:192], yb)


@pytest.mark.parametrize("t,hw,heads", [(16, 36, 5), (4, 64, 2), (5, 7, 3), (20, 9, 1)])
def test_temporal_attention_and_groupnorm_do_not_depend_on_the_batch(cuda, t, hw, heads):
    """A clip's result must carry the same bits whether its pass runs alone or stacked with other clips (the samplers stack the guidance
    passes; clip-level data parallelism splits them): the MFMA temporal attention and the register-table GroupNorm, stacked vs one by one."""
    from mudg_amd import ops
    c = heads * 64
    qkv = rnd(3 * t * hw, 3 * c, seed=5).to(cuda)
    out = torch.zeros(3 * t * hw, c, dtype=BF, device=cuda)
    ops.temporal_attention(qkv, out, clips=3, t=t, hw=hw, heads=heads)
    for b in range(3):
        one = torch.zeros(t * hw, c, dtype=BF, device=cuda)
        ops.temporal_attention(qkv[b * t * hw:(b + 1) * t * hw].contiguous(), one, clips=1, t=t, hw=hw, heads=heads)
        assert torch.equal(one, out[b * t * hw:(b + 1) * t * hw])
    rows, cg = 150, 960
    x = (rnd(3 * rows, cg, seed=6).float() * 2 + 0.3).to(BF).to(cuda)
    g = (1 + 0.1 * torch.randn(cg, generator=torch.Generator().manual_seed(2))).to(cuda)
    bb = (0.1 * torch.randn(cg, generator=torch.Generator().manual_seed(3))).to(cuda)
    y = ops.groupnorm(x, g, bb, samples=3, rows=rows, eps=1e-5, silu=True)
    for b in range(3):
        one = ops.groupnorm(x[b * rows:(b + 1) * rows].contiguous(), g, bb, samples=1, rows=rows, eps=1e-5, silu=True)
        assert torch.equal(one, y[b * rows:(b + 1) * rows])


@pytest.mark.parametrize("samples,rows,c,silu,eps", [(2, 50, 64, True, 1e-5), (3, 144, 320, True, 1e-5),
                                                     (1, 700, 960, False, 1e-6), (2, 33, 2560, True, 1e-5),
                                                     (1, 5000, 128, True, 1e-6), (2, 20, 32, True, 1e-5)])
def test_groupnorm(cuda, samples, rows, c, silu, eps):
    from mudg_amd import ops
    x = (rnd(samples * rows, c, seed=1).float() * 2 + 0.7).to(BF)
    g = 1 + 0.1 * torch.randn(c, generator=torch.Generator().manual_seed(2))
    b = 0.1 * torch.randn(c, generator=torch.Generator().manual_seed(3))
    xr = x.float().reshape(samples, rows, c).transpose(1, 2)               # (N, C, L)
    ref = F.group_norm(xr, 32, g, b, eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.transpose(1, 2).reshape(samples * rows, c)
    y = ops.groupnorm(x.to(cuda), g.to(cuda), b.to(cuda), samples=samples, rows=rows, eps=eps, silu=silu)
    assert rel_l2(y, ref) < TOL_BF16


def test_groupnorm_two_sources(cuda):
    from mudg_amd import ops
    samples, rows, c1, c2 = 2, 40, 320, 640
    x1, x2 = rnd(samples * rows, c1, seed=1), rnd(samples * rows, c2, seed=2, scale=3.0)
    g = 1 + 0.1 * torch.randn(c1 + c2, generator=torch.Generator().manual_seed(3))
    b = 0.1 * torch.randn(c1 + c2, generator=torch.Generator().manual_seed(4))
    x = torch.cat([x1, x2], 1).float().reshape(samples, rows, -1).transpose(1, 2)
    ref = F.silu(F.group_norm(x, 32, g, b, 1e-5)).transpose(1, 2).reshape(samples * rows, -1)
    y = ops.groupnorm(x1.to(cuda), g.to(cuda), b.to(cuda), samples=samples, rows=rows, eps=1e-5, silu=True,
                      x2=x2.to(cuda))
    assert rel_l2(y, ref) < TOL_BF16


@pytest.mark.parametrize("rows,c", [(10, 320), (77, 1280), (5, 2048), (3, 64), (9, 512), (33, 640), (7, 1024), (130, 320)])
def test_layernorm(cuda, rows, c):
    from mudg_amd import ops
    x = (rnd(rows, c, seed=1).float() * 3 - 1).to(BF)
    g = 1 + 0.1 * torch.randn(c, generator=torch.Generator().manual_seed(2))
    b = 0.1 * torch.randn(c, generator=torch.Generator().manual_seed(3))
    ref = F.layer_norm(x.float(), (c,), g, b, 1e-5)
    y = ops.layernorm(x.to(cuda), g.to(cuda), b.to(cuda))
    assert rel_l2(y, ref) < TOL_BF16


def test_softmax_rows(cuda):
    from mudg_amd import ops
    s = torch.randn(37, 1000, generator=torch.Generator().manual_seed(1)) * 4
    y = ops.softmax_rows(s.to(cuda))
    assert rel_l2(y, torch.softmax(s, -1)) < TOL_BF16


def test_timestep_embedding_known_answers(cuda):
    """SURVEY Appendix C values captured from the reference's timestep_embedding([999,19,10,500], 320)."""
    from mudg_amd import ops
    e = ops.timestep_embedding(torch.tensor([999, 19, 10, 500], device=cuda), 320).cpu()
    assert torch.allclose(e[0, :3], torch.tensor([0.9996498, 0.8026775, -0.27806213]), atol=2e-5)
    assert torch.allclose(e[0, 160:163], torch.tensor([-0.026460752, 0.5964133, -0.9605631]), atol=2e-5)
    assert torch.allclose(e.sum(1), torch.tensor([57.183205, 125.40709, 136.76776, 69.293144]), atol=2e-3)


def test_small_linear(cuda):
    from mudg_amd import ops
    g = torch.Generator().manual_seed(1)
    x, w, b = torch.randn(3, 320, generator=g), torch.randn(1280, 320, generator=g) * 0.05, torch.randn(1280, generator=g)
    y = ops.small_linear(x.to(cuda), w.to(cuda), b.to(cuda), act_out=True)
    assert rel_l2(y, F.silu(x @ w.t() + b)) < TOL_F32
    y2 = ops.small_linear(x.to(cuda), w.to(cuda), b.to(cuda), act_in=True)
    assert rel_l2(y2, F.silu(x) @ w.t() + b) < TOL_F32


def test_layout_round_trip(cuda):
    from mudg_amd import ops
    x = torch.randn(2, 4, 3, 5, 6, generator=torch.Generator().manual_seed(1))
    c = torch.randn(2, 8, 3, 5, 6, generator=torch.Generator().manual_seed(2))
    rows = torch.full((2 * 3 * 30, 16), 7.0, dtype=BF, device=cuda)
    ops.ncthw_to_rows(x.to(cuda), rows, 0)
    ops.ncthw_to_rows(c.to(cuda), rows, 4)
    ops.zero_channels(rows, 12, 16)
    want = torch.cat([x, c, torch.zeros(2, 4, 3, 5, 6)], 1).permute(0, 2, 3, 4, 1).reshape(-1, 16).to(BF)
    assert torch.equal(rows.cpu(), want)
    back = ops.rows_to_ncthw(rows, (2, 4, 3, 5, 6), coff=4)
    assert torch.equal(back.cpu(), c[:, :4].to(BF).float())


@pytest.mark.parametrize("phi,with_uncond,with_noise", [(0.7, True, True), (0.0, True, False), (0.7, False, True)])
def test_ddim_step(cuda, phi, with_uncond, with_noise):
    from mudg_amd import ops
    g = torch.Generator().manual_seed(1)
    shape = (3, 4, 16, 9, 16)
    x, ec, eu, nz = (torch.randn(shape, generator=g) for _ in range(4))
    ec = ec * torch.tensor([1.0, 2.0, 0.5]).view(3, 1, 1, 1, 1)
    cfg, sac, s1m, resc, sap, dirc, sig = 7.5, 0.6, 0.8, 1.03, 0.9, 0.3, 0.31
    v = eu + cfg * (ec - eu) if with_uncond else ec
    if phi > 0:
        dims = list(range(1, v.ndim))
        v = phi * (v * (ec.std(dim=dims, keepdim=True) / v.std(dim=dims, keepdim=True))) + (1 - phi) * v
    e = sac * v + s1m * x
    x0 = (sac * x - s1m * v) * resc
    xp = sap * x0 + dirc * e + (sig * nz if with_noise else 0)
    got_xp, got_x0 = ops.ddim_step(x.to(cuda), ec.to(cuda), eu.to(cuda) if with_uncond else None,
                                   nz.to(cuda) if with_noise else None, [cfg, phi, sac, s1m, resc, sap, dirc, sig])
    assert rel_l2(got_x0, x0) < TOL_F32 and rel_l2(got_xp, xp) < TOL_F32


def test_fp32_residual_stream_variants(cuda):
    """The residual stream is fp32: GroupNorm / LayerNorm read it in fp32, GEMM epilogues add an fp32 residual and
    write fp32; the fp32 -> bf16 cast feeds raw-stream GEMM operands."""
    from mudg_amd import ops
    g = torch.Generator().manual_seed(11)
    samples, rows, c = 2, 60, 320
    x = torch.randn(samples * rows, c, generator=g) * 2 + 0.3
    gam, bet = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    ref = F.silu(F.group_norm(x.reshape(samples, rows, c).transpose(1, 2), 32, gam, bet, 1e-5)).transpose(1, 2).reshape(-1, c)
    y = ops.groupnorm(x.to(cuda), gam.to(cuda), bet.to(cuda), samples=samples, rows=rows, eps=1e-5, silu=True)
    assert y.dtype == BF and rel_l2(y, ref) < TOL_BF16
    x2 = torch.randn(samples * rows, 64, generator=g)
    gam2, bet2 = torch.cat([gam, gam[:64]]), torch.cat([bet, bet[:64]])
    ref2 = F.group_norm(torch.cat([x, x2], 1).reshape(samples, rows, -1).transpose(1, 2), 32, gam2, bet2, 1e-6)
    y2 = ops.groupnorm(x.to(cuda), gam2.to(cuda), bet2.to(cuda), samples=samples, rows=rows, eps=1e-6, silu=False,
                       x2=x2.to(cuda))
    assert rel_l2(y2, ref2.transpose(1, 2).reshape(-1, c + 64)) < TOL_BF16
    yl = ops.layernorm(x.to(cuda), gam.to(cuda), bet.to(cuda))
    assert rel_l2(yl, F.layer_norm(x, (c,), gam, bet, 1e-5)) < TOL_BF16
    a, w = rnd(samples * rows, 64, seed=1), rnd(c, 64, seed=2, scale=0.1)
    out = ops.gemm(a.to(cuda), w.to(cuda), residual=x.to(cuda), out_fp32=True)
    assert out.dtype == torch.float32 and rel_l2(out, a.float() @ w.float().t() + x) < TOL_F32
    xb = ops.cast_bf16(x.to(cuda))
    assert torch.equal(xb.cpu(), x.to(BF))
    odd = torch.randn(1003, generator=g)
    assert torch.equal(ops.cast_bf16(odd.to(cuda)).cpu(), odd.to(BF))


# ---- large problems (many tiles: the high-occupancy single-buffer kernels and, where it applies, the phase-scheduled large tile)
def test_large_gemm_plain_epilogues(cuda):
    from mudg_amd import ops
    M, N, K = 4096 + 40, 3072, 200                      # ragged M, K tail (200 = 3 tiles + 8)
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3))
    r32 = torch.randn(M, N, generator=torch.Generator().manual_seed(4))
    ref = x.float() @ w.float().t() + b
    y = ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda))
    assert rel_l2(y, ref) < TOL_BF16
    y32 = ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda), residual=r32.to(cuda), out_fp32=True)
    assert rel_l2(y32, ref + r32) < TOL_F32
    rb = rnd(M, N, seed=5)
    gb = torch.randn(M // 8 + 1, N, generator=torch.Generator().manual_seed(6))
    y3 = ops.gemm(x.to(cuda), w.to(cuda), residual=rb.to(cuda), gbias=gb.to(cuda), rows_per_group=8)
    assert rel_l2(y3, x.float() @ w.float().t() + rb.float() + gb.repeat_interleave(8, 0)[:M]) < TOL_BF16


def test_large_gemm_transpose_detecting_and_two_sources(cuda):
    from mudg_amd import ops
    n = 4096
    w = ((torch.arange(n * 256).reshape(n, 256) * 7) % 251).float().to(BF)      # asymmetric
    x = torch.zeros(n, 256)
    x[torch.arange(256) * 16, torch.arange(256)] = 1.0                          # row 16 j selects column j
    y = ops.gemm(x.to(BF).to(cuda), w.to(cuda), out_fp32=True)                  # [4096, 4096]
    want = torch.zeros(n, n)
    want[torch.arange(256) * 16] = w.float().t()
    assert torch.equal(y.cpu(), want)
    M, N, K1, K2 = 4096, 3072, 64, 192
    x1, x2, w2 = rnd(M, K1, seed=1), rnd(M, K2, seed=2), rnd(N, K1 + K2, seed=3, scale=0.05)
    y2 = ops.gemm(x1.to(cuda), w2.to(cuda), x2=x2.to(cuda))
    assert rel_l2(y2, torch.cat([x1, x2], 1).float() @ w2.float().t()) < TOL_BF16


def test_large_gemm_geglu(cuda):
    from mudg_amd import ops
    M, C = 8192, 192                                      # N = 8C = 1536: 32 x 6 = 192 tiles
    x, w = rnd(M, C, seed=1), rnd(8 * C, C, seed=2, scale=0.1)
    b = torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1
    val, gate = (x.float() @ w.float().t() + b).chunk(2, dim=-1)
    wp, bp = pack_geglu(w, b)
    y = ops.gemm(x.to(cuda), wp.to(cuda), bias=bp.to(cuda), geglu=True)
    assert tuple(y.shape) == (M, 4 * C) and rel_l2(y, val * F.gelu(gate)) < TOL_BF16


@pytest.mark.parametrize("korder,stride,ups", [(0, 1, False), (1, 1, False), (1, 2, False), (1, 1, True)])
def test_large_gemm_conv3x3(cuda, korder, stride, ups):
    from mudg_amd import ops
    frames, h, w, cin, cout = (16, 32, 32, 64, 768) if not ups else (4, 32, 32, 64, 768)
    if stride == 2:
        frames, h, w = 16, 64, 64
    x = rnd(frames, cin, h, w, seed=1)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=0.05)
    b = torch.randn(cout, generator=torch.Generator().manual_seed(3))
    xin = F.interpolate(x.float(), scale_factor=2, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, wt.float(), b, stride=stride, padding=1)
    packed = pack_conv_slab(wt) if korder else pack_conv(wt)
    y = ops.conv3x3(to_rows(x).to(cuda), packed.to(cuda), frames=frames, hin=h, win=w, cin=cin, stride=stride,
                    upsample=ups, bias=b.to(cuda), korder=korder)
    assert rel_l2(from_rows(y.cpu().float(), frames, ref.shape[2], ref.shape[3]), ref) < TOL_BF16


def test_large_gemm_tconv3(cuda):
    from mudg_amd import ops
    clips, t, h, w, c = 2, 8, 32, 32, 256                 # M = 16384, N = 256 -> 64 tiles: forced below
    x = rnd(clips, c, t, h, w, seed=1)
    wt = rnd(3 * c, c, 3, 1, 1, seed=2, scale=0.05)        # Cout = 768 so that 64 x 3 = 192 tiles
    ref = F.conv3d(x.float(), wt.float(), None, padding=(1, 0, 0))
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, c).contiguous()
    wp = wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(3 * c, 3 * c).contiguous()
    y = ops.tconv3(rows.to(cuda), wp.to(cuda), clips=clips, t=t, hw=h * w, cin=c)
    got = y.cpu().float().reshape(clips, t, h, w, 3 * c).permute(0, 4, 1, 2, 3)
    assert rel_l2(got, ref) < TOL_BF16


@pytest.mark.parametrize("clips,hw,c,cout", [(2, 1024, 128, 192), (1, 72, 64, 64), (3, 2304, 320, 320)])
def test_tconv3_slab_major_tiles_of_8_pixels_by_16_frames(cuda, clips, hw, c, cout):
    """ops.tconv3(korder=1): a tile is 8 pixels x 16 frames of a clip, the three taps share one staged slab (one-stage kernels) — the
    same convolution (vs torch, vs the tap-major kernel: equal sums in another K order), residual and bias included; its GroupNorm
    partials serve a clip-level norm, and a frame-level norm must NOT take them (it falls back to its own statistics pass)."""
    from mudg_amd import hip, ops
    if not ops.tconv3_slab_ok(16, hw, c):
        pytest.skip("korder 1 belongs to the 16-bit and bf16x3 builds")
    t = 16
    x = rnd(clips, c, t, hw, 1, seed=1)
    wt = rnd(cout, c, 3, 1, 1, seed=2, scale=0.05)
    b = torch.randn(cout, generator=torch.Generator().manual_seed(3))
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, c).contiguous()
    res = rnd(clips * t * hw, cout, seed=4)
    ref = F.conv3d(x.float(), wt.float(), b, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(-1, cout) + res.float()
    w_tap = wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(cout, 3 * c).contiguous()
    w_slab = wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(cout, 3, c // 64, 64).permute(0, 2, 1, 3).reshape(cout, 3 * c).contiguous()
    kw = dict(clips=clips, t=t, hw=hw, cin=c, bias=b.to(cuda), residual=res.to(cuda), stats=True)
    y0 = ops.tconv3(rows.to(cuda), w_tap.to(cuda), **kw)
    y1 = ops.tconv3(rows.to(cuda), w_slab.to(cuda), korder=1, **kw)
    assert rel_l2(y1, ref) < TOL_BF16 and rel_l2(y1, y0.float().cpu()) < 2e-3
    gam, bet = torch.ones(cout, device=cuda), torch.zeros(cout, device=cuda)
    if (t * hw) % 128 == 0:
        fused = ops.groupnorm(y1, gam, bet, samples=clips, rows=t * hw, eps=1e-5, silu=True)          # the tiles' partials
        plain = ops.groupnorm(y1, gam, bet, samples=clips, rows=t * hw, eps=1e-5, silu=True, fused=False)
        assert rel_l2(fused, plain.float().cpu()) < 5e-4
    frame = ops.groupnorm(y1, gam, bet, samples=clips * t, rows=hw, eps=1e-5, silu=False)               # must not use them
    fref = F.group_norm(y1.float().cpu().reshape(clips * t, hw, cout).transpose(1, 2), 32, None, None, 1e-5).transpose(1, 2).reshape(-1, cout)
    assert rel_l2(frame, fref) < TOL_BF16


def test_ddim_step_three_way_guidance(cuda):
    from mudg_amd import ops
    g = torch.Generator().manual_seed(2)
    shape = (2, 4, 8, 9, 16)
    x, ec, eu, em, nz = (torch.randn(shape, generator=g) for _ in range(5))
    cfg, cimg, phi, sac, s1m, resc, sap, dirc, sig = 7.5, 2.0, 0.7, 0.6, 0.8, 1.03, 0.9, 0.3, 0.31
    v = eu + cimg * (em - eu) + cfg * (ec - em)
    dims = list(range(1, v.ndim))
    v = phi * (v * (ec.std(dim=dims, keepdim=True) / v.std(dim=dims, keepdim=True))) + (1 - phi) * v
    x0 = (sac * x - s1m * v) * resc
    xp = sap * x0 + dirc * (sac * v + s1m * x) + sig * nz
    got_xp, got_x0 = ops.ddim_step(x.to(cuda), ec.to(cuda), eu.to(cuda), nz.to(cuda),
                                   [cfg, phi, sac, s1m, resc, sap, dirc, sig, cimg], e_m=em.to(cuda))
    assert rel_l2(got_x0, x0) < TOL_F32 and rel_l2(got_xp, xp) < TOL_F32


# ---- GroupNorm statistics fused into the producing epilogue (MudgGemmDesc.stats -> mudg_groupnorm_fused)
@pytest.mark.parametrize("out_fp32", [False, True])
def test_groupnorm_fused_partials_match_the_two_pass_norm(cuda, out_fp32):
    from mudg_amd import ops
    g = torch.Generator().manual_seed(11)
    frames, h, w, cin, c1, c2 = 6, 16, 16, 64, 320, 64               # rows per frame = 256 = 2 partial blocks
    x = rnd(frames * h * w, cin, seed=1)
    w1, w2 = rnd(c1, 9 * cin, seed=2, scale=0.05), rnd(c2, cin, seed=3, scale=0.1)
    res = torch.randn(frames * h * w, c1, generator=g)
    a = ops.conv3x3(x.to(cuda), w1.to(cuda), frames=frames, hin=h, win=w, cin=cin, residual=res.to(cuda) if out_fp32 else None,
                    out_fp32=out_fp32, stats=True)
    b = ops.gemm(x.to(cuda), w2.to(cuda), out_fp32=out_fp32, stats=True)
    ra, rb = getattr(a, ops.GN_ATTR + "_rows"), getattr(b, ops.GN_ATTR + "_rows")      # 128; 288 where the variant child forces the 288 x 320 tile
    assert getattr(a, ops.GN_ATTR).shape == (-(-frames * h * w // ra), c1, 2) and getattr(b, ops.GN_ATTR).shape == (-(-frames * h * w // rb), c2, 2)
    gam = (1 + 0.1 * torch.randn(c1 + c2, generator=g)).to(cuda)
    bet = (0.1 * torch.randn(c1 + c2, generator=g)).to(cuda)
    for samples, rows in ((frames, h * w), (2, 3 * h * w)):          # per-frame and per-clip statistics from the same partials
        fused = ops.groupnorm(a, gam, bet, samples=samples, rows=rows, eps=1e-5, silu=True, x2=b)
        plain = ops.groupnorm(a, gam, bet, samples=samples, rows=rows, eps=1e-5, silu=True, x2=b, fused=False)
        ref = F.silu(F.group_norm(torch.cat([a, b], 1).float().cpu().reshape(samples, rows, -1).transpose(1, 2), 32,
                                  gam.cpu(), bet.cpu(), 1e-5)).transpose(1, 2).reshape(-1, c1 + c2)
        assert rel_l2(fused, plain.float().cpu()) < 5e-4                   # bf16 outputs of (nearly) the same statistics
        assert rel_l2(fused, ref) < TOL_BF16
    one = ops.groupnorm(a, gam[:c1].contiguous(), bet[:c1].contiguous(), samples=frames, rows=h * w, eps=1e-6, silu=False)
    one_ref = F.group_norm(a.float().cpu().reshape(frames, h * w, c1).transpose(1, 2), 32, gam[:c1].cpu(), bet[:c1].cpu(), 1e-6)
    assert rel_l2(one, one_ref.transpose(1, 2).reshape(-1, c1)) < TOL_BF16


def test_groupnorm_fused_from_tconv_and_large_tiles(cuda):
    from mudg_amd import ops
    clips, t, hw, c = 2, 8, 1024, 256                                   # M = 16384, N = 768: the 256-wide kernels when forced
    x = rnd(clips * t * hw, c, seed=1)
    wp = rnd(3 * c, 3 * c, seed=2, scale=0.03)
    y = ops.tconv3(x.to(cuda), wp.to(cuda), clips=clips, t=t, hw=hw, cin=c, stats=True)
    gam, bet = torch.ones(3 * c, device=cuda), torch.zeros(3 * c, device=cuda)
    fused = ops.groupnorm(y, gam, bet, samples=clips, rows=t * hw, eps=1e-5, silu=True)
    plain = ops.groupnorm(y, gam, bet, samples=clips, rows=t * hw, eps=1e-5, silu=True, fused=False)
    assert rel_l2(fused, plain.float().cpu()) < 5e-4
    # a sample that is not a whole number of 128-row blocks takes the two-pass path (rows = 1000 here)
    part = y[:16000]
    setattr(part, ops.GN_ATTR, getattr(y, ops.GN_ATTR))
    setattr(part, ops.GN_ATTR + "_version", part._version)
    odd = ops.groupnorm(part, gam, bet, samples=16, rows=1000, eps=1e-5, silu=False)
    odd_ref = F.group_norm(part.float().cpu().reshape(16, 1000, 3 * c).transpose(1, 2), 32, None, None, 1e-5)
    assert rel_l2(odd, odd_ref.transpose(1, 2).reshape(-1, 3 * c)) < TOL_BF16
    # a torch in-place update after production invalidates the partials: the norm must see the new values
    y[::2].mul_(3.0)              # every other row: not a per-group affine map, so stale statistics would show
    upd = ops.groupnorm(y, gam, bet, samples=clips, rows=t * hw, eps=1e-5, silu=False)
    upd_ref = F.group_norm(y.float().cpu().reshape(clips, t * hw, 3 * c).transpose(1, 2), 32, None, None, 1e-5)
    assert rel_l2(upd, upd_ref.transpose(1, 2).reshape(-1, 3 * c)) < TOL_BF16


# ---- the wide tiles (256 x 320 / 256 x 256, eight waves): chosen by mudg_gemm for problems with many tiles whose N is a
# multiple of 320 (or 256 with GEGLU); the variant child GEMM_WIDE=1 (tests/test_gemm_variants_gpu.py) forces them on every
# shape here, the last test reaches them through the library's own selection rule.
@pytest.mark.parametrize("M,N,K", [(1000, 320, 320), (515, 640, 64), (256, 960, 1280), (70, 320, 128)])
def test_wide_tile_plain_epilogues_ragged_rows(cuda, M, N, K):
    from mudg_amd import ops
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3))
    r = rnd(M, N, seed=4)
    ref = x.float() @ w.float().t() + b
    assert rel_l2(ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda), residual=r.to(cuda)), ref + r.float()) < TOL_BF16
    r32 = torch.randn(M, N, generator=torch.Generator().manual_seed(5))
    y32 = ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda), residual=r32.to(cuda), out_fp32=True, alpha=0.5)
    assert rel_l2(y32, 0.5 * (x.float() @ w.float().t()) + b + r32) < TOL_F32
    gb = torch.randn((M + 6) // 7, N, generator=torch.Generator().manual_seed(6))       # groups of 7 rows straddle every tile
    y3 = ops.gemm(x.to(cuda), w.to(cuda), gbias=gb.to(cuda), rows_per_group=7, out_fp32=True)
    assert rel_l2(y3, x.float() @ w.float().t() + gb.repeat_interleave(7, 0)[:M]) < TOL_F32


def test_wide_tile_transpose_detecting(cuda):
    from mudg_amd import ops
    n, k = 640, 256
    w = ((torch.arange(n * k).reshape(n, k) * 7) % 251).float().to(BF)      # asymmetric
    x = torch.zeros(512, k)
    x[torch.arange(k) * 2, torch.arange(k)] = 1.0                          # row 2 j selects column j
    y = ops.gemm(x.to(BF).to(cuda), w.to(cuda), out_fp32=True)
    want = torch.zeros(512, n)
    want[torch.arange(k) * 2] = w.float().t()
    assert torch.equal(y.cpu(), want)


def test_wide_tile_geglu_and_two_sources(cuda):
    from mudg_amd import ops
    M, C = 700, 128                                       # N = 8C = 1024: the 256-wide GEGLU tile
    x, w = rnd(M, C, seed=1), rnd(8 * C, C, seed=2, scale=0.1)
    b = torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1
    val, gate = (x.float() @ w.float().t() + b).chunk(2, dim=-1)
    wp, bp = pack_geglu(w, b)
    y = ops.gemm(x.to(cuda), wp.to(cuda), bias=bp.to(cuda), geglu=True)
    assert tuple(y.shape) == (M, 4 * C) and rel_l2(y, val * F.gelu(gate)) < TOL_BF16
    x1, x2, w2 = rnd(M, 64, seed=4), rnd(M, 192, seed=5), rnd(320, 256, seed=6, scale=0.05)
    y2 = ops.gemm(x1.to(cuda), w2.to(cuda), x2=x2.to(cuda))
    assert rel_l2(y2, torch.cat([x1, x2], 1).float() @ w2.float().t()) < TOL_BF16


@pytest.mark.parametrize("stride,two,stats", [(1, False, True), (2, False, False), (1, True, True)])
def test_wide_tile_conv3x3_with_fused_epilogue_and_partials(cuda, stride, two, stats):
    from mudg_amd import ops
    frames, h, w, cin, cout = 5, 16, 24, 128, 320                         # 384 rows per frame = 3 partial blocks of 128
    x = rnd(frames, cin, h, w, seed=1)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=0.05)
    b = torch.randn(cout, generator=torch.Generator().manual_seed(3))
    ref = F.conv2d(x.float(), wt.float(), b, stride=stride, padding=1)
    ho, wo = ref.shape[2], ref.shape[3]
    emb = torch.randn(frames, cout, generator=torch.Generator().manual_seed(5))
    res = rnd(frames, cout, ho, wo, seed=6)
    ref = ref + emb[:, :, None, None] + res.float()
    rows = to_rows(x)
    kw = dict(x2=rows[:, 64:].to(cuda)) if two else {}
    xin = rows[:, :64].contiguous() if two else rows
    y = ops.conv3x3(xin.to(cuda), pack_conv_slab(wt).to(cuda), frames=frames, hin=h, win=w, cin=cin, stride=stride, korder=1,
                    bias=b.to(cuda), gbias=emb.to(cuda), rows_per_group=ho * wo, residual=to_rows(res).to(cuda), stats=stats, **kw)
    assert rel_l2(from_rows(y.cpu().float(), frames, ho, wo), ref) < TOL_BF16
    if stats:
        gam, bet = torch.ones(cout, device=cuda), torch.zeros(cout, device=cuda)
        fused = ops.groupnorm(y, gam, bet, samples=frames, rows=ho * wo, eps=1e-5, silu=True)
        plain = ops.groupnorm(y, gam, bet, samples=frames, rows=ho * wo, eps=1e-5, silu=True, fused=False)
        assert rel_l2(fused, plain.float().cpu()) < 5e-4


def test_wide_tile_tconv3_and_subpixel_upsample(cuda):
    from mudg_amd import ops
    from mudg_amd.engine import packing
    clips, t, h, w, c = 2, 6, 8, 10, 320
    x = rnd(clips, c, t, h, w, seed=1)
    wt = rnd(c, c, 3, 1, 1, seed=2, scale=0.05)
    b = torch.randn(c, generator=torch.Generator().manual_seed(3))
    ref = F.conv3d(x.float(), wt.float(), b, padding=(1, 0, 0)) + x.float()
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, c).contiguous()
    wp = wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, 3 * c).contiguous()
    y = ops.tconv3(rows.to(cuda), wp.to(cuda), clips=clips, t=t, hw=h * w, cin=c, bias=b.to(cuda), residual=rows.to(cuda))
    assert rel_l2(y.cpu().float().reshape(clips, t, h, w, c).permute(0, 4, 1, 2, 3), ref) < TOL_BF16
    # nearest-2x upsample + 3x3 conv in the sub-pixel form (four parity classes = batch 4)
    conv = torch.nn.Conv2d(64, 320, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(rnd(320, 64, 3, 3, seed=7, scale=0.05).float())
        conv.bias.copy_(torch.randn(320, generator=torch.Generator().manual_seed(8)))
    conv = conv.to(cuda)
    xs = rnd(3, 64, 9, 12, seed=9)
    wsub = packing.conv3x3_subpixel(conv)
    if wsub is not None:
        out = ops.conv3x3_up2(to_rows(xs).to(cuda), wsub, frames=3, hin=9, win=12, cin=64, bias=packing.f32(conv, "bias"))
        if out is not None:
            want = F.conv2d(F.interpolate(xs.float(), scale_factor=2, mode="nearest"), conv.weight.detach().cpu().to(BF).float(),
                            conv.bias.detach().cpu(), padding=1)
            assert rel_l2(from_rows(out.cpu().float(), 3, 18, 24), want) < 2 * TOL_BF16


def test_wide_tile_is_selected_for_many_tile_problems_and_matches_the_small_tile(cuda):
    """M = 200 000 rows x N = 320: 782 wide tiles -> the library picks the 256 x 320 kernel by itself; the same rows computed as
    four quarter problems (196 tiles each: the 128 x 128 kernels) must agree BIT FOR BIT — every output element accumulates
    its K-tiles in the same order through the same MFMA in both kernels (the premise of clip-independent results)."""
    from mudg_amd import ops
    M, N, K = 200000, 320, 192
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3))
    xd, wd, bd = x.to(cuda), w.to(cuda), b.to(cuda)
    whole = ops.gemm(xd, wd, bias=bd)
    q = M // 4
    parts = torch.cat([ops.gemm(xd[i * q:(i + 1) * q], wd, bias=bd) for i in range(4)], 0)
    assert torch.equal(whole, parts)
    assert rel_l2(whole[::97], x[::97].float() @ w.float().t() + b) < TOL_BF16


# ---- the persistent kernel (pgemm.hip): every GEGLU problem and plain GEMMs with N, K >= 1280 under the library's rule; the
# variant children GEMM_PERSIST=4 / 3 / 2 / 0 (tests/test_gemm_variants_gpu.py) force it on (or off for) every FAST problem of
# this file.  These tests reach it through the rule, with more tiles than persistent workgroups (each walks several tiles).
def test_persistent_geglu_walks_many_tiles_and_rows_do_not_depend_on_the_batch(cuda):
    from mudg_amd import ops
    M, C = 20000 + 40, 320                                # 157 row tiles (the last ragged) x 20 column tiles = 3140 tiles > 1024 workgroups
    x, w = rnd(M, C, seed=1), rnd(8 * C, C, seed=2, scale=0.06)
    b = torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1
    val, gate = (x.float() @ w.float().t() + b).chunk(2, dim=-1)
    wp, bp = pack_geglu(w, b)
    xd, wd, bd = x.to(cuda), wp.to(cuda), bp.to(cuda)
    y = ops.gemm(xd, wd, bias=bd, geglu=True)
    assert tuple(y.shape) == (M, 4 * C) and rel_l2(y, val * F.gelu(gate)) < TOL_BF16
    part = ops.gemm(xd[:4096 + 128], wd, bias=bd, geglu=True)          # fewer tiles: another walk, the same arithmetic per tile
    assert torch.equal(part, y[:4096 + 128])
    again = ops.gemm(xd, wd, bias=bd, geglu=True)
    assert torch.equal(again, y)


@pytest.mark.parametrize("kind", ["operand", "stream", "f32"])
def test_persistent_plain_gemm_with_residual_seed_and_groupnorm_partials(cuda, kind):
    """N, K >= 1280: the persistent kernel under the library's rule.  The residual seeds the accumulators (every storage kind
    of it), the epilogue writes GroupNorm partials of what it stored, and a clip's rows do not depend on what else is in M."""
    from mudg_amd import ops
    M, N, K = 128 * 150 + 72, 1280, 1280                   # 151 x 10 = 1510 tiles on 768 (3 per CU) workgroups, ragged last row tile
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.03)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3)) * 0.1
    r = torch.randn(M, N, generator=torch.Generator().manual_seed(4))
    res = {"operand": r.to(BF), "stream": r.to(ops.STREAM()), "f32": r}[kind]
    ref = x.float() @ w.float().t() + b + res.float()
    kw = dict(out_stream=kind == "stream", out_fp32=kind == "f32")
    y = ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda), residual=res.to(cuda), stats=True, **kw)
    assert rel_l2(y, ref) < (TOL_F32 if kind == "f32" else TOL_BF16)
    stats = getattr(y, ops.GN_ATTR)                        # [ceil(M / rows)][N][2]: sum and sum of squares of the stored values
    rows = getattr(y, ops.GN_ATTR + "_rows")               # 128 (288 in the variant child that forces the 288 x 320 tile)
    assert tuple(stats.shape) == ((M + rows - 1) // rows, N, 2)
    yf = torch.nn.functional.pad(y.float().cpu(), (0, 0, 0, (-M) % rows)).reshape(-1, rows, N)
    assert rel_l2(stats[..., 0], yf.sum(1)) < 1e-5 and rel_l2(stats[..., 1], (yf * yf).sum(1)) < 1e-5
    sub = 5760                                             # whole blocks of every height (45 x 128 = 36 x 160 = 20 x 288)
    part = ops.gemm(x[:sub].to(cuda), w.to(cuda), bias=b.to(cuda), residual=res[:sub].to(cuda), stats=True, **kw)
    assert torch.equal(part, y[:sub]) and torch.equal(getattr(part, ops.GN_ATTR), stats[:sub // rows])


def test_persistent_rule_takes_ragged_widths_and_scaled_residuals_elsewhere(cuda):
    """What the persistent kernel does not take (a width that is not a multiple of 8, alpha != 1 with a residual, the plain GELU)
    still runs — on the one-tile kernels — and stays correct at N, K >= 1280."""
    from mudg_amd import ops
    M, N, K = 1024, 1284, 1280
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.03)
    r = torch.randn(M, N, generator=torch.Generator().manual_seed(4))
    y = ops.gemm(x.to(cuda), w.to(cuda), residual=r.to(cuda), out_fp32=True)
    assert rel_l2(y, x.float() @ w.float().t() + r) < TOL_F32
    w8 = w[:1280].contiguous()
    y2 = ops.gemm(x.to(cuda), w8.to(cuda), residual=r[:, :1280].contiguous().to(cuda), out_fp32=True, alpha=0.5)
    assert rel_l2(y2, 0.5 * (x.float() @ w8.float().t()) + r[:, :1280]) < TOL_F32
    y3 = ops.gemm(x.to(cuda), w8.to(cuda), gelu=True, out_fp32=True)
    assert rel_l2(y3, F.gelu(x.float() @ w8.float().t())) < 1e-4


# ---------------------------------------------------------------------------------------------- the 288 x 320 tile (csrc/wgemm.hip)
def _conv_ref(x, w, frames, h, wd, cin, cout, korder):
    """fp32 conv2d of the bf16-rounded inputs; w is the packed [Cout][9 cin] matrix in tap-major (0) or slab-major (1) K order."""
    if korder:
        wk = w.float().reshape(cout, cin // 64, 9, 64).permute(0, 2, 1, 3).reshape(cout, 3, 3, cin)
    else:
        wk = w.float().reshape(cout, 3, 3, cin)
    xi = x.float().reshape(frames, h, wd, cin).permute(0, 3, 1, 2)
    return F.conv2d(xi, wk.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(frames * h * wd, cout)


def _wide_build():
    return _hip.planes() <= 2           # the kernel belongs to the 16-bit builds and bf16x3 (this file runs in the 16-bit builds)


def _rows_ok(rows, by_rule):
    """The partial-block height the library reported against what its RULE gives for the case; a variant child (debug-variants build) may
    have forced another tile kernel or switched one off: then any of the three heights is the library's to choose."""
    import os
    if os.environ.get("MUDG_DEBUG_VARIANTS") == "1":
        return rows in (128, 160, 288)
    return rows == by_rule


@pytest.mark.parametrize("kind,korder", [("operand", 1), ("stream", 0), ("f32", 1)])
def test_wide288_conv_group_bias_residual_two_sources_and_partials(cuda, kind, korder):
    """Frames of 576 = 2 x 288 pixels: the library's rule sends the conv to the 288 x 320 tile (16-bit builds).  Bias + per-frame
    group bias, a residual of every storage kind, the input split over two channel sources, GroupNorm partials per 288-row block."""
    from mudg_amd import ops
    f, h, wd, c1, c2, cout = 3, 24, 24, 64, 128, 320
    cin, M = c1 + c2, 3 * 24 * 24
    xa, xb = rnd(M, c1, seed=1), rnd(M, c2, seed=2)
    w = rnd(cout, 9 * cin, seed=3, scale=0.03)
    b = torch.randn(cout, generator=torch.Generator().manual_seed(4)) * 0.1
    gb = torch.randn(f, cout, generator=torch.Generator().manual_seed(5))
    r = torch.randn(M, cout, generator=torch.Generator().manual_seed(6))
    res = {"operand": r.to(BF), "stream": r.to(ops.STREAM()), "f32": r}[kind]
    ref = _conv_ref(torch.cat([xa, xb], 1), w, f, h, wd, cin, cout, korder) + b + gb.repeat_interleave(h * wd, 0) + res.float()
    kw = dict(out_stream=kind == "stream", out_fp32=kind == "f32")
    y = ops.conv3x3(xa.to(cuda), w.to(cuda), x2=xb.to(cuda), frames=f, hin=h, win=wd, cin=cin, korder=korder, bias=b.to(cuda), gbias=gb.to(cuda),
                    rows_per_group=h * wd, residual=res.to(cuda), stats=True, **kw)
    assert rel_l2(y, ref) < (TOL_F32 if kind == "f32" else TOL_BF16)
    rows = getattr(y, ops.GN_ATTR + "_rows")
    assert _rows_ok(rows, 288 if _wide_build() else 128)
    stats = getattr(y, ops.GN_ATTR)
    assert tuple(stats.shape) == (M // rows if M % rows == 0 else M // rows + 1, cout, 2)
    if M % rows == 0:
        yf = y.float().cpu().reshape(-1, rows, cout)
        assert rel_l2(stats[..., 0], yf.sum(1)) < 1e-5 and rel_l2(stats[..., 1], (yf * yf).sum(1)) < 1e-5
    # a frame's rows and partial blocks do not depend on what else is in the launch
    one = ops.conv3x3(xa[:h * wd].to(cuda), w.to(cuda), x2=xb[:h * wd].to(cuda), frames=1, hin=h, win=wd, cin=cin, korder=korder, bias=b.to(cuda),
                      gbias=gb[:1].to(cuda), rows_per_group=h * wd, residual=res[:h * wd].to(cuda), stats=True, **kw)
    assert torch.equal(one, y[:h * wd]) and torch.equal(getattr(one, ops.GN_ATTR), stats[:h * wd // rows])


def test_wide288_plain_gemm_ragged_rows_and_temporal_conv(cuda):
    from mudg_amd import ops
    M, N, K = 288 * 5 + 100, 320, 1280                    # K >= 1280 at N = 320 with a frame hint of whole tiles: the rule's plain-GEMM case
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.03)
    b = torch.randn(N, generator=torch.Generator().manual_seed(3)) * 0.1
    r = rnd(M, N, seed=4).to(ops.STREAM())
    ref = x.float() @ w.float().t() + b + r.float()
    y = ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda), residual=r.to(cuda), out_stream=True, stats=True, frame_rows=288)
    assert rel_l2(y, ref) < TOL_BF16
    rows = getattr(y, ops.GN_ATTR + "_rows")
    assert _rows_ok(rows, 288 if _wide_build() else 128) and getattr(y, ops.GN_ATTR).shape[0] == (M + rows - 1) // rows
    # no hint: the 128 x 128 kernels, the same bits (compared without the residual: the persistent 128 x 128 kernel, which a variant child
    # forces, adds a residual first instead of last)
    assert torch.equal(ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda), out_fp32=True), ops.gemm(x.to(cuda), w.to(cuda), bias=b.to(cuda), out_fp32=True, frame_rows=288))
    # temporal conv (plain K order), N = 640: clips of 4 frames x 288 pixels; the first and last frame of a clip see zero padding
    clips, t, hw, c, co = 2, 4, 288, 128, 640
    xt, wt = rnd(clips * t * hw, c, seed=5), rnd(co, 3 * c, seed=6, scale=0.05)
    bt = torch.randn(co, generator=torch.Generator().manual_seed(7)) * 0.1
    xi = xt.float().reshape(clips, t, hw, c).permute(0, 3, 1, 2)                                            # (clip, c, t, hw)
    wk = wt.float().reshape(co, 3, c).permute(0, 2, 1)                                                     # (co, c, tap)
    ref = F.conv2d(xi, wk[..., None], padding=(1, 0)).permute(0, 2, 3, 1).reshape(clips * t * hw, co) + bt
    assert ops.tconv3_wide(t, hw, c, co) == _wide_build()
    yt = ops.tconv3(xt.to(cuda), wt.to(cuda), clips=clips, t=t, hw=hw, cin=c, bias=bt.to(cuda), stats=True)
    assert rel_l2(yt, ref) < TOL_BF16
    rows = getattr(yt, ops.GN_ATTR + "_rows")
    yf = yt.float().cpu().reshape(-1, rows, co)
    assert rel_l2(getattr(yt, ops.GN_ATTR)[..., 0], yf.sum(1)) < 1e-5


def test_wide288_partials_feed_the_groupnorm_also_next_to_128_row_partials(cuda):
    """GroupNorm over two channel sources whose producers wrote partial blocks of different heights (a 288 x 320-tile conv and a
    128 x 128-tile GEMM): the fused norm folds each source's blocks on their own; against the norm that takes its own statistics."""
    from mudg_amd import ops
    f, h, wd, cin = 2, 24, 48, 64                          # 1152 = 4 x 288 = 9 x 128 rows per frame
    M = f * h * wd
    x, w1 = rnd(M, cin, seed=1), rnd(320, 9 * cin, seed=2, scale=0.05)
    a = ops.conv3x3(x.to(cuda), w1.to(cuda), frames=f, hin=h, win=wd, cin=cin, korder=1, stats=True, out_stream=True)
    bsrc = ops.gemm(x.to(cuda), rnd(160, cin, seed=3, scale=0.1).to(cuda), stats=True, out_stream=True)     # N = 160: the 128 x 128 kernels
    if _wide_build():
        assert _rows_ok(getattr(a, ops.GN_ATTR + "_rows"), 288) and getattr(bsrc, ops.GN_ATTR + "_rows") == 128
    g = torch.randn(480, generator=torch.Generator().manual_seed(4)).to(cuda)
    bt = torch.randn(480, generator=torch.Generator().manual_seed(5)).to(cuda)
    for x1, x2 in ((a, None), (a, bsrc), (bsrc, a)):
        c = x1.shape[1] + (x2.shape[1] if x2 is not None else 0)
        fused = ops.groupnorm(x1, g[:c], bt[:c], samples=f, rows=h * wd, eps=1e-5, silu=True, groups=32, x2=x2)
        plain = ops.groupnorm(x1, g[:c], bt[:c], samples=f, rows=h * wd, eps=1e-5, silu=True, groups=32, x2=x2, fused=False)
        assert rel_l2(fused, plain) < 2e-3                 # (both round to operands; the statistics agree to fp32 rounding)
        xs = torch.cat([x1.float()] + ([x2.float()] if x2 is not None else []), 1).cpu().reshape(f, h * wd, 32, c // 32)
        mean, var = xs.mean((1, 3), keepdim=True), xs.var((1, 3), unbiased=False, keepdim=True)
        ref = F.silu(((xs - mean) / torch.sqrt(var + 1e-5)).reshape(M, c) * g[:c].cpu() + bt[:c].cpu())
        assert rel_l2(fused, ref) < TOL_BF16


def test_wide288_persistent_geglu_walks_a_stream_of_tiles(cuda):
    """GEGLU at K = 640 with frames of whole 288-row tiles: the library's rule sends it to the 288 x 256 tile, in the persistent form when
    there are more tiles than CUs (here 40 x 20 = 800: a workgroup walks three or four tiles, the K-tiles of which form one stream through
    its LDS ring).  Against the fp32 reference; a sub-batch (fewer tiles than CUs: the one-tile form) gives the same bits."""
    from mudg_amd import ops
    M, C = 288 * 40, 640
    x, w = rnd(M, C, seed=1), rnd(8 * C, C, seed=2, scale=0.04)
    b = torch.randn(8 * C, generator=torch.Generator().manual_seed(3)) * 0.1
    val, gate = (x.float() @ w.float().t() + b).chunk(2, dim=-1)
    wp, bp = pack_geglu(w, b)
    xd, wd, bd = x.to(cuda), wp.to(cuda), bp.to(cuda)
    y = ops.gemm(xd, wd, bias=bd, geglu=True, frame_rows=288 * 2)
    assert tuple(y.shape) == (M, 4 * C) and rel_l2(y, val * F.gelu(gate)) < TOL_BF16
    part = ops.gemm(xd[:288 * 6], wd, bias=bd, geglu=True, frame_rows=288 * 2)            # 6 x 20 = 120 tiles: one per workgroup
    assert torch.equal(part, y[:288 * 6])
    assert torch.equal(ops.gemm(xd, wd, bias=bd, geglu=True, frame_rows=288 * 2), y)
    y32 = ops.gemm(xd, wd, bias=bd, geglu=True, out_fp32=True, frame_rows=288 * 2)
    assert rel_l2(y32, val * F.gelu(gate)) < 1e-4


def test_wide288_is_bit_identical_to_the_one_tile_kernels(cuda):
    """Debug-variants build only (the switch is read at every call there): the same problem on the 128 x 128 kernels and on the
    288 x 320 tile gives the same bits — the K-tile order is the same and v_mfma_f32_16x16x32 adds its 32 products as two
    v_mfma_f32_32x32x16 add their 16 each.  (Without a residual: the tile takes one as its accumulators' initial value, the one-tile
    128 x 128 kernels add it last.)"""
    import os
    if os.environ.get("MUDG_DEBUG_VARIANTS") != "1" or os.environ.get("MUDG_GEMM_PERSIST", "1") not in ("0", "1"):
        pytest.skip("needs the debug-variants build with the default kernel rule (tests/test_gemm_variants_gpu.py runs it)")
    from mudg_amd import ops
    f, h, wd, cin, cout = 2, 24, 36, 128, 640
    x, w = rnd(f * h * wd, cin, seed=1).to(cuda), rnd(cout, 9 * cin, seed=2, scale=0.03).to(cuda)
    r = rnd(f * h * wd, cout, seed=3).to(ops.STREAM()).to(cuda)
    xm, wm = rnd(288 * 7 + 31, 640, seed=4).to(cuda), rnd(960, 640, seed=5, scale=0.03).to(cuda)
    xl = rnd(288 * 300, 640, seed=6).to(cuda)              # 900 tiles: with GEMM_W288P=2 the persistent form
    outs = []
    saved = os.environ.get("MUDG_GEMM_W288")
    try:
        for v in ("0", "2"):
            os.environ["MUDG_GEMM_W288"] = v
            outs.append((ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=1, out_stream=True, stats=True),
                         ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=0, out_fp32=True),
                         ops.gemm(xm, wm, out_fp32=True), ops.tconv3(x, w[:, :3 * cin].contiguous(), clips=1, t=2, hw=h * wd, cin=cin),
                         ops.gemm(xl, wm, out_stream=True, stats=True, frame_rows=288)))
            res_out = ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=1, residual=r, out_stream=True)
            if v == "0":
                res_ref = res_out
            else:       # a residual seeds the 288 x 320 tile's accumulators, the one-tile 128 x 128 kernels add it last: fp32 rounding apart
                assert rel_l2(res_out, res_ref.float().cpu()) < 2e-3 and float((res_out.float() - res_ref.float()).abs().max()) < 0.05
    finally:
        if saved is None:
            os.environ.pop("MUDG_GEMM_W288", None)
        else:
            os.environ["MUDG_GEMM_W288"] = saved
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # the same tile on the loop of the 160-row kernel (wq_kernel<..., 9>, GEMM_W288Q = 2: a measurement kept in the variant builds): the
    # same bits as the six-phase loop, a residual included (both seed the accumulators with it)
    if _hip.planes() == 1:
        saved = {k: os.environ.get(k) for k in ("MUDG_GEMM_W288", "MUDG_GEMM_W288Q")}
        try:
            pair = []
            for q in ("0", "2"):
                os.environ["MUDG_GEMM_W288"], os.environ["MUDG_GEMM_W288Q"] = "2", q
                pair.append((ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=1, residual=r, out_stream=True, stats=True),
                             ops.gemm(xm, wm, out_fp32=True), ops.tconv3(x, w[:, :3 * cin].contiguous(), clips=1, t=2, hw=h * wd, cin=cin),
                             ops.gemm(xl[:288 * 9], wm, residual=rnd(288 * 9, 960, seed=9).to(ops.STREAM()).to(cuda), out_stream=True, stats=True, frame_rows=288)))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        for a, b in zip(*pair):
            assert torch.equal(a, b) and torch.equal(getattr(a, ops.GN_ATTR, torch.zeros(1)), getattr(b, ops.GN_ATTR, torch.zeros(1)))


# ---------------------------------------------------------------------------------------------- the 160 x 320 tile (csrc/wgemm.hip: w160_kernel)
def _w160_build():
    return _hip.planes() == 1            # 16-bit builds only


def test_wide160_conv_gemm_and_temporal_conv_by_the_rule(cuda):
    """Frames of 20 x 32 = 640 pixels (MDM512's second level: whole 160-row tiles, no whole 288-row tiles): the library's rule sends the 3x3
    conv, the long-K plain GEMM and the temporal conv to the 160 x 320 tile (16-bit builds).  Conv with bias + per-frame group bias, an
    fp16-stream residual, two channel sources and GroupNorm partials per 160-row block; GEMM with ragged rows; against fp32 references;
    a frame's rows and partial blocks do not depend on what else is in the launch."""
    from mudg_amd import ops
    f, h, wd, c1, c2, cout = 3, 20, 32, 64, 128, 320
    cin, M = c1 + c2, 3 * 20 * 32
    xa, xb = rnd(M, c1, seed=1), rnd(M, c2, seed=2)
    w = rnd(cout, 9 * cin, seed=3, scale=0.03)
    b = torch.randn(cout, generator=torch.Generator().manual_seed(4)) * 0.1
    gb = torch.randn(f, cout, generator=torch.Generator().manual_seed(5))
    res = torch.randn(M, cout, generator=torch.Generator().manual_seed(6)).to(ops.STREAM())
    ref = _conv_ref(torch.cat([xa, xb], 1), w, f, h, wd, cin, cout, 1) + b + gb.repeat_interleave(h * wd, 0) + res.float()
    y = ops.conv3x3(xa.to(cuda), w.to(cuda), x2=xb.to(cuda), frames=f, hin=h, win=wd, cin=cin, korder=1, bias=b.to(cuda), gbias=gb.to(cuda),
                    rows_per_group=h * wd, residual=res.to(cuda), stats=True, out_stream=True)
    assert rel_l2(y, ref) < TOL_BF16
    rows = getattr(y, ops.GN_ATTR + "_rows")
    assert _rows_ok(rows, 160 if _w160_build() else 128)
    stats = getattr(y, ops.GN_ATTR)
    blocks = lambda t, r: torch.nn.functional.pad(t.float().cpu(), (0, 0, 0, (-t.shape[0]) % r)).reshape(-1, r, t.shape[1])     # (a variant child may force another height)
    yf = blocks(y, rows)
    assert tuple(stats.shape) == ((M + rows - 1) // rows, cout, 2)
    assert rel_l2(stats[..., 0], yf.sum(1)) < 1e-5 and rel_l2(stats[..., 1], (yf * yf).sum(1)) < 1e-5
    one = ops.conv3x3(xa[:h * wd].to(cuda), w.to(cuda), x2=xb[:h * wd].to(cuda), frames=1, hin=h, win=wd, cin=cin, korder=1, bias=b.to(cuda),
                      gbias=gb[:1].to(cuda), rows_per_group=h * wd, residual=res[:h * wd].to(cuda), stats=True, out_stream=True)
    assert torch.equal(one, y[:h * wd]) and ((h * wd) % rows or torch.equal(getattr(one, ops.GN_ATTR), stats[:h * wd // rows]))
    # the partials feed the fused GroupNorm (160-row blocks next to a 128-row source)
    other = ops.gemm(xa.to(cuda), rnd(160, c1, seed=7, scale=0.1).to(cuda), stats=True, out_stream=True)       # N = 160: the 128 x 128 kernels
    g = torch.randn(480, generator=torch.Generator().manual_seed(8)).to(cuda)
    bt = torch.randn(480, generator=torch.Generator().manual_seed(9)).to(cuda)
    fused = ops.groupnorm(y, g, bt, samples=f, rows=h * wd, eps=1e-5, silu=True, groups=32, x2=other)
    plain = ops.groupnorm(y, g, bt, samples=f, rows=h * wd, eps=1e-5, silu=True, groups=32, x2=other, fused=False)
    assert rel_l2(fused, plain.float().cpu()) < 2e-3
    # plain GEMM, K = 1280, ragged rows, frame hint 640; without the hint the 128 x 128 kernels: the same bits without a residual
    Mg, N, K = 160 * 9 + 100, 320, 1280
    x, wg = rnd(Mg, K, seed=11), rnd(N, K, seed=12, scale=0.03)
    bg = torch.randn(N, generator=torch.Generator().manual_seed(13)) * 0.1
    r = rnd(Mg, N, seed=14).to(ops.STREAM())
    yg = ops.gemm(x.to(cuda), wg.to(cuda), bias=bg.to(cuda), residual=r.to(cuda), out_stream=True, stats=True, frame_rows=640)
    assert rel_l2(yg, x.float() @ wg.float().t() + bg + r.float()) < TOL_BF16
    rows = getattr(yg, ops.GN_ATTR + "_rows")
    assert _rows_ok(rows, 160 if _w160_build() else 128) and getattr(yg, ops.GN_ATTR).shape[0] == (Mg + rows - 1) // rows
    assert torch.equal(ops.gemm(x.to(cuda), wg.to(cuda), bias=bg.to(cuda), out_fp32=True), ops.gemm(x.to(cuda), wg.to(cuda), bias=bg.to(cuda), out_fp32=True, frame_rows=640))
    # temporal conv (plain K order): clips of 4 frames x 640 pixels
    clips, t, hw, c, co = 2, 4, 640, 128, 640
    xt, wt = rnd(clips * t * hw, c, seed=15), rnd(co, 3 * c, seed=16, scale=0.05)
    btc = torch.randn(co, generator=torch.Generator().manual_seed(17)) * 0.1
    xi = xt.float().reshape(clips, t, hw, c).permute(0, 3, 1, 2)
    wk = wt.float().reshape(co, 3, c).permute(0, 2, 1)
    ref = F.conv2d(xi, wk[..., None], padding=(1, 0)).permute(0, 2, 3, 1).reshape(clips * t * hw, co) + btc
    assert ops.tconv3_wide(t, hw, c, co) == _w160_build() or os.environ.get("MUDG_DEBUG_VARIANTS") == "1"
    yt = ops.tconv3(xt.to(cuda), wt.to(cuda), clips=clips, t=t, hw=hw, cin=c, bias=btc.to(cuda), stats=True)
    assert rel_l2(yt, ref) < TOL_BF16
    rows = getattr(yt, ops.GN_ATTR + "_rows")
    assert rel_l2(getattr(yt, ops.GN_ATTR)[..., 0], blocks(yt, rows).sum(1)) < 1e-5


def test_wide160_is_bit_identical_to_the_one_tile_kernels(cuda):
    """Debug-variants build only: the same problem on the 128 x 128 kernels and on the 160 x 320 tile forced for everything it can run
    (GEGLU on its 160 x 256 form, K from one K-tile — shorter than the ring — up, ragged M, one-tile frames whose every row touches an image
    border) gives the same bits without a residual; with one the tile adds it first (fp32 rounding apart); repeated launches reproduce."""
    import os
    if os.environ.get("MUDG_DEBUG_VARIANTS") != "1" or os.environ.get("MUDG_GEMM_PERSIST", "1") not in ("0", "1"):
        pytest.skip("needs the debug-variants build with the default kernel rule (tests/test_gemm_variants_gpu.py runs it)")
    from mudg_amd import hip, ops
    if hip.planes() != 1:
        pytest.skip("the 16-bit builds' kernel")
    f, h, wd, cin, cout = 2, 10, 16, 128, 640
    x, w = rnd(f * h * wd, cin, seed=1).to(cuda), rnd(cout, 9 * cin, seed=2, scale=0.03).to(cuda)
    r = rnd(f * h * wd, cout, seed=3).to(ops.STREAM()).to(cuda)
    xm, wm = rnd(160 * 7 + 31, 640, seed=4).to(cuda), rnd(960, 640, seed=5, scale=0.03).to(cuda)
    xs, ws = rnd(160 * 3 + 1, 64, seed=6).to(cuda), rnd(320, 64, seed=7, scale=0.1).to(cuda)          # one K-tile: two k halves, a ring of five
    xk, wk = rnd(500, 192, seed=8).to(cuda), rnd(320, 192, seed=9, scale=0.1).to(cuda)                # three K-tiles: no steady-state iteration
    xg, wg, bg = rnd(160 * 5 + 7, 320, seed=10).to(cuda), rnd(1024, 320, seed=11, scale=0.1).to(cuda), torch.randn(1024, device=cuda)
    outs = []
    saved = {k: os.environ.get(k) for k in ("MUDG_GEMM_W288", "MUDG_GEMM_W160")}
    try:
        for v in ("0", "2", "2"):
            os.environ["MUDG_GEMM_W288"], os.environ["MUDG_GEMM_W160"] = "0", v
            outs.append((ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=1, out_stream=True, stats=True),
                         ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=0, out_fp32=True),
                         ops.gemm(xm, wm, out_fp32=True), ops.tconv3(x, w[:, :3 * cin].contiguous(), clips=1, t=2, hw=h * wd, cin=cin),
                         ops.gemm(xs, ws, out_stream=True), ops.gemm(xk, wk), ops.gemm(xg, wg, bias=bg, geglu=True), ops.gemm(xg, wg, bias=bg, geglu=True, out_fp32=True)))
            rows = getattr(outs[-1][0], ops.GN_ATTR + "_rows")
            assert rows == (128 if v == "0" else 160)
            res_out = ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=1, residual=r, out_stream=True)
            r32_out = ops.conv3x3(x, w, frames=f, hin=h, win=wd, cin=cin, korder=1, residual=r.float(), out_stream=True)
            if v == "0":
                res_ref, r32_ref = res_out, r32_out
            else:
                # a 16-bit residual is deferred to the tile's epilogue — ((x w + bias) + r), the one-tile kernels' order: the same bits;
                # an fp32 one seeds the accumulators: fp32 rounding apart
                assert torch.equal(res_out, res_ref)
                assert rel_l2(r32_out, r32_ref.float().cpu()) < 2e-3 and float((r32_out.float() - r32_ref.float()).abs().max()) < 0.05
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for a, b, c in zip(*outs):
        assert torch.isfinite(a.float()).all() and float(a.float().abs().max()) > 1e-3
        assert torch.equal(a, b) and torch.equal(b, c)


def test_half_height_geglu_kernel_is_bit_identical_to_the_kernels_it_replaces(cuda):
    """Debug-variants build only: the two-workgroup 144 x 256 GEGLU kernel of round 6 (csrc/wgemm.hip: hgeglu_kernel, both K-loop forms)
    against the persistent / one-tile 288 x 256 tile and the 128 x 128 kernels on the same inputs — same K order, the same Phi table
    arithmetic: the same bits, which is what lets its selection rule look at M.  Ragged M (a last tile of 13 rows, a single row), two
    activation sources, K from one K-tile up, fp32 and operand results."""
    import os
    if os.environ.get("MUDG_DEBUG_VARIANTS") != "1" or os.environ.get("MUDG_GEMM_PERSIST", "1") not in ("0", "1"):
        pytest.skip("needs the debug-variants build with the default kernel rule (tests/test_gemm_variants_gpu.py runs it)")
    from mudg_amd import hip, ops
    if hip.planes() != 1:
        pytest.skip("the 16-bit builds' kernel")
    cases = [(144 * 5, 512, 320, False, 0), (144 * 7 + 13, 256, 64, False, 0), (1, 256, 128, True, 0), (288 * 40 + 100, 2560, 320, False, 0),
             (144 * 33, 1024, 1280, True, 0), (5000, 768, 192, False, 64)]
    saved = {k: os.environ.get(k) for k in ("MUDG_GEMM_W288", "MUDG_GEMM_H144", "MUDG_GEMM_H144PF")}
    try:
        for i, (M, N, K, f32, split) in enumerate(cases):
            x, w, b = rnd(M, K - split, seed=10 + i).to(cuda), rnd(N, K, seed=20 + i, scale=0.1).to(cuda), torch.randn(N, device=cuda)
            x2 = rnd(M, split, seed=30 + i).to(cuda) if split else None
            outs = []
            for w288, h144, pf in (("0", "0", "1"), ("2", "0", "1"), ("2", "2", "1"), ("2", "2", "0")):
                os.environ.update(MUDG_GEMM_W288=w288, MUDG_GEMM_H144=h144, MUDG_GEMM_H144PF=pf)
                outs.append(ops.gemm(x, w, x2=x2, bias=b, geglu=True, out_fp32=f32, frame_rows=288))
            torch.cuda.synchronize()
            assert torch.isfinite(outs[0].float()).all() and float(outs[0].float().abs().max()) > 1e-3
            for o in outs[1:]:
                assert torch.equal(o, outs[0]), (M, N, K)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_zero_initialised_alpha_is_one_for_the_kernel_choice_and_the_groupnorm_blocks(cuda):
    """Round-5 advisor finding, on the GPU: a descriptor with alpha = 0 (a zero-initialised C struct: mudg_gemm reads it as 1), a residual
    and GroupNorm partials.  mudg_gemm_stats_rows used to answer for the un-normalised descriptor (128-row blocks) while mudg_gemm ran
    the 288 x 320 tile and wrote 288-row blocks.  Now both see the same descriptor: the block height the query reports is the one the
    partials were written in (they equal the column sums of the stored result taken in blocks of that height), and the result is the
    alpha = 1 result bit for bit."""
    from mudg_amd import ops
    M, N, K = 288 * 6, 320, 320
    x, w = rnd(M, K, seed=1).to(cuda), rnd(N, K, seed=2, scale=0.05).to(cuda)
    b = torch.randn(N, device=cuda)
    r = rnd(M, N, seed=3).to(ops.STREAM()).to(cuda)
    outs = []
    for alpha in (0.0, 1.0):
        y = ops.gemm(x, w, bias=b, residual=r, stats=True, out_stream=True, alpha=alpha, frame_rows=288)
        outs.append((y, getattr(y, ops.GN_ATTR), getattr(y, ops.GN_ATTR + "_rows")))
    torch.cuda.synchronize()
    (y0, p0, r0), (y1, p1, r1) = outs
    assert r0 == r1 and torch.equal(y0, y1) and torch.equal(p0, p1)
    v = y0.double().reshape(M // r0, r0, N)
    want = torch.stack([v.sum(1), (v ** 2).sum(1)], -1)
    assert p0.shape == (M // r0, N, 2)
    assert rel_l2(p0, want.cpu()) < 1e-5

