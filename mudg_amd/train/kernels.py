"""Tensor-level wrappers over the training entries of the C-ABI (csrc/train.hip).  fp32 rows matrices in, fp32 rows (or MFMA
operand matrices) out; nothing here computes."""
import ctypes as C

import torch

from .. import hip, ops


def _s():
    return torch.cuda.current_stream().cuda_stream


def _f32(t):
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
        raise hip.MudgError(f"expected a cuda fp32 rows matrix, got {tuple(t.shape)} {t.dtype} {t.stride()} on {t.device}")
    return t


def transpose_gather(src, P=None, mode=0, geo=None, out=None, batch=1, src_batch_rows=0, dst_batch_rows=0):
    """dst[c][p] = src[srcrow(p)][c] as an operand matrix [C][P rounded up to 8] (mudg_transpose_gather).  batch > 1: entry z
    transposes the P rows starting at row z * src_batch_rows of `src` into the C rows starting at row z * dst_batch_rows of `out`."""
    _f32(src)
    P = src.shape[0] if P is None else P
    Cc = src.shape[1]
    Ppad = (P + 7) // 8 * 8
    if out is None:
        out = ops.empty_rows(Cc, Ppad, ops.H16(), src.device)
    g = dict(Hin=0, Win=0, Hout=0, Wout=0, stride=1, pad=1, dy=0, dx=0, T=0, HW=0, dt=0)
    g.update(geo or {})
    hip.check(hip.lib().mudg_transpose_gather(src.data_ptr(), src.stride(0), out.data_ptr(), out.stride(0), P, Cc, mode, g["Hin"], g["Win"],
                                              g["Hout"], g["Wout"], g["stride"], g["pad"], g["dy"], g["dx"], g["T"], g["HW"], g["dt"], batch,
                                              src_batch_rows * src.stride(0), dst_batch_rows * out.stride(0), _s()),
              "mudg_transpose_gather")
    return out


def _colsum_once(a, b, rpg):
    rows, cols = a.shape
    out = torch.empty((rows // rpg, cols), dtype=torch.float32, device=a.device)
    hip.check(hip.lib().mudg_group_colsum(a.data_ptr(), a.stride(0), None if b is None else _f32(b).data_ptr(), 0 if b is None else b.stride(0),
                                          rows, cols, rpg, out.data_ptr(), _s()), "mudg_group_colsum")
    return out


def group_colsum(a, b=None, rows_per_group=None):
    """out[g][c] = sum over the rows of group g of a[r][c] (* b[r][c]).  One launch gives a workgroup per (64 columns, group): a
    group of many rows is first cut into chunks (a divisor of its length, <= 1024 rows), whose partial sums are then summed —
    two launches on many workgroups instead of one on a handful; the chunking depends on the shape only (fixed order)."""
    _f32(a)
    rows = a.shape[0]
    rpg = rows if rows_per_group is None else rows_per_group
    if rpg > 2048:
        for chunk in (1024, 768, 640, 576, 512, 384, 320, 256, 192, 160, 128, 96, 64):
            if rpg % chunk == 0:
                return _colsum_once(_colsum_once(a, b, chunk), None, rpg // chunk)
    return _colsum_once(a, b, rpg)


def groupnorm_stats(x, samples, rows, groups, eps):
    _f32(x)
    stat = torch.empty((samples * groups, 2), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().mudg_groupnorm_stats(x.data_ptr(), x.stride(0), samples, rows, x.shape[1], groups, eps, stat.data_ptr(), _s()),
              "mudg_groupnorm_stats")
    return stat


def groupnorm_bwd(x, dy, gamma, beta, stat, samples, rows, groups, silu, dres=None):
    """dres: fp32 rows added to dx (the gradient of the residual branch around the norm)."""
    _f32(x); _f32(dy)
    c = x.shape[1]
    dx = torch.empty((samples * rows, c), dtype=torch.float32, device=x.device)
    ab = torch.empty((samples, 2 * c), dtype=torch.float32, device=x.device)
    ws = torch.empty(hip.lib().mudg_groupnorm_bwd_ws_floats(samples, rows, c, groups), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().mudg_groupnorm_bwd(x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                           stat.data_ptr(), samples, rows, c, groups, int(silu), dx.data_ptr(), dx.stride(0), ab.data_ptr(),
                                           ws.data_ptr(), None if dres is None else _f32(dres).data_ptr(), 0 if dres is None else dres.stride(0),
                                           _s()), "mudg_groupnorm_bwd")
    tot = group_colsum(ab)[0]                       # sums over the samples: [(dbeta_c, dgamma_c) interleaved]
    return dx, tot[1::2].contiguous(), tot[0::2].contiguous()


def layernorm_bwd(x, dy, gamma, eps, dres=None):
    """dres: fp32 rows added to dx (the gradient of the residual branch around the norm)."""
    _f32(x); _f32(dy)
    rows, c = x.shape
    dx = torch.empty_like(x)
    chunks = hip.lib().mudg_layernorm_bwd_chunks(rows)
    part = torch.empty((chunks, 2 * c), dtype=torch.float32, device=x.device)           # per chunk of rows: [sum dy xhat | sum dy]
    hip.check(hip.lib().mudg_layernorm_bwd(x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), gamma.data_ptr(), dx.data_ptr(), dx.stride(0),
                                           part.data_ptr(), rows, c, eps, None if dres is None else _f32(dres).data_ptr(),
                                           0 if dres is None else dres.stride(0), _s()), "mudg_layernorm_bwd")
    sums = group_colsum(part)[0]
    return dx, sums[:c], sums[c:]


def transpose_cast_sum(src, out, rows=False, sums=False):
    """mudg_transpose_cast_sum: one pass over fp32 rows `src` [P][C] writing out[c][p] (operand matrix, the caller's; None: no
    transposed copy), and on request the operand-row copy and the column sums.  Returns (rows or None, sums or None)."""
    _f32(src)
    P, c = src.shape
    r = ops.empty_rows(P, c, ops.H16(), src.device) if rows else None
    part = torch.empty(((P + 63) // 64, c), dtype=torch.float32, device=src.device) if sums else None
    hip.check(hip.lib().mudg_transpose_cast_sum(src.data_ptr(), src.stride(0), None if out is None else out.data_ptr(),
                                                0 if out is None else out.stride(0), None if r is None else r.data_ptr(),
                                                0 if r is None else r.stride(0), None if part is None else part.data_ptr(), P, c, _s()),
              "mudg_transpose_cast_sum")
    return r, (group_colsum(part)[0] if sums else None)


def geglu(h, dy=None):
    _f32(h)
    m, n2 = h.shape
    n = n2 // 2
    out = torch.empty((m, n if dy is None else n2), dtype=torch.float32, device=h.device)
    hip.check(hip.lib().mudg_geglu(h.data_ptr(), h.stride(0), None if dy is None else _f32(dy).data_ptr(), 0 if dy is None else dy.stride(0),
                                   out.data_ptr(), out.stride(0), m, n, _s()), "mudg_geglu")
    return out


def dropout_rows(x, p, seed, operand=False):
    """mudg_dropout_rows on fp32 rows (the mask of `dropout` on the same contiguous rows) -> (fp32 rows, operand rows or None)."""
    _f32(x)
    m, c = x.shape
    y = torch.empty((m, c), dtype=torch.float32, device=x.device)
    y16 = ops.empty_rows(m, c, ops.H16(), x.device) if operand else None
    hip.check(hip.lib().mudg_dropout_rows(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), None if y16 is None else y16.data_ptr(),
                                          0 if y16 is None else y16.stride(0), m, c, float(p), int(seed), _s()), "mudg_dropout_rows")
    return y, y16


def geglu_dropout(h, p, seed, dy=None, operand=False):
    """mudg_geglu_dropout: forward (dy None) -> (fp32 rows [M][N], operand rows or None); backward -> dH [M][2 N]."""
    _f32(h)
    m, n2 = h.shape
    n = n2 // 2
    out = torch.empty((m, n if dy is None else n2), dtype=torch.float32, device=h.device)
    o16 = ops.empty_rows(m, n, ops.H16(), h.device) if (operand and dy is None) else None
    hip.check(hip.lib().mudg_geglu_dropout(h.data_ptr(), h.stride(0), None if dy is None else _f32(dy).data_ptr(), 0 if dy is None else dy.stride(0),
                                           out.data_ptr(), out.stride(0), None if o16 is None else o16.data_ptr(), 0 if o16 is None else o16.stride(0),
                                           m, n, float(p), int(seed), _s()), "mudg_geglu_dropout")
    return (out, o16) if dy is None else out


def softmax_f32(s, cols):
    """In place over the first `cols` columns of every row."""
    hip.check(hip.lib().mudg_softmax_f32(s.data_ptr(), s.stride(0), s.data_ptr(), s.stride(0), s.shape[0], cols, _s()), "mudg_softmax_f32")
    return s


def softmax_bwd(p, dp, ds, cols, scale):
    hip.check(hip.lib().mudg_softmax_bwd(p.data_ptr(), p.stride(0), dp.data_ptr(), dp.stride(0), ds.data_ptr(), ds.stride(0), p.shape[0], cols,
                                         scale, _s()), "mudg_softmax_bwd")
    return ds


def attention_bwd(q, k, v, do, *, frames, heads, nq, nk, kv_div, scale, o=None, lse=None, out=None):
    """mudg_attention_bwd: (dq, dk, dv) fp32 rows of softmax(scale q k^T) v from operand rows q, do [frames * nq][C], k, v
    [(frames / kv_div) * nk][C] (16-bit operand builds).  out: (dq, dk, dv) fp32 row views to write into (dk and dv with one row
    stride), e.g. the column blocks of one packed gradient."""
    dev = q.device
    c = heads * 64
    # o / lse: the forward output and the statistics it saved (ops.attention(lse=...)): no statistics pass
    stat = torch.empty((2, frames * nq, heads), dtype=torch.float32, device=dev)
    big_l = stat[0] if lse is None else lse
    if out is not None:
        dq, dk, dv = out
        if dk.stride(0) != dv.stride(0) or any(t.dtype != torch.float32 or t.stride(1) != 1 for t in out):
            raise hip.MudgError("attention_bwd: out must be fp32 row views, dk and dv with the same row stride")
    else:
        dq = torch.empty((frames * nq, c), dtype=torch.float32, device=dev)
        dk = torch.empty((frames // kv_div * nk, c), dtype=torch.float32, device=dev)
        dv = torch.empty_like(dk)
    d = hip.AttnBwdDesc()
    d.Q, d.K, d.V, d.dO = q.data_ptr(), k.data_ptr(), v.data_ptr(), do.data_ptr()
    d.L, d.D = big_l.data_ptr(), stat[1].data_ptr()
    d.dQ, d.dK, d.dV = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    d.F, d.heads, d.Nq, d.Nk, d.kv_div = frames, heads, nq, nk, kv_div
    d.ldq, d.ldk, d.ldv, d.lddo = q.stride(0), k.stride(0), v.stride(0), do.stride(0)
    if o is not None and lse is not None:
        d.O, d.ldo = o.data_ptr(), o.stride(0)
    d.ldgq, d.ldgk = dq.stride(0), dk.stride(0)
    d.scale = scale
    hip.check(hip.lib().mudg_attention_bwd(C.byref(d), _s()), "mudg_attention_bwd")
    return dq, dk, dv


def wgrad(a, b, *, positions, m, c, taps=1, mode=0, geo=None):
    """mudg_wgrad: out[m][tap * c + ch] = sum_p a[p][m] b[src(p, tap)][ch] from operand rows a [positions][>= m], b [*][>= c] (16-bit
    operand builds; c % 64 == 0, m % 8 == 0) -> fp32 [m][taps * c].  The contraction is cut into slices sized from the problem
    shape only (enough workgroups for the chip), whose slabs are added in a fixed order."""
    n = taps * c
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    # 512 workgroups run at a time (2 per CU).  A slice count k costs rounds(k) / k passes over the positions at ~1 TFLOP/s per
    # workgroup, plus writing and re-reading k fp32 slabs at ~4 TB/s: take the cheapest (a function of the shape only).
    most = max(1, min(64, positions // 1024))
    tile_us = 2.0 * 128 * 128 * positions / 1e6
    slab_us = m * n * 8 / 4e6
    slices = min(range(1, most + 1), key=lambda k: (-(-tiles * k // 512) / k * tile_us + (k * slab_us if k > 1 else 0.0), k))
    chunk = ((positions + slices - 1) // slices + 63) // 64 * 64
    slices = (positions + chunk - 1) // chunk
    slabs = torch.empty((slices, m * n), dtype=torch.float32, device=a.device)
    g = dict(Hin=0, Win=0, Hout=0, Wout=0, stride=1, pad=1, T=0, HW=0)
    g.update(geo or {})
    d = hip.WgradDesc()
    d.A, d.B, d.out = a.data_ptr(), b.data_ptr(), slabs.data_ptr()
    d.lda, d.ldb, d.P = a.stride(0), b.stride(0), positions
    d.M, d.C, d.taps, d.mode = m, c, taps, mode
    d.Hin, d.Win, d.Hout, d.Wout, d.stride, d.pad, d.T, d.HW = g["Hin"], g["Win"], g["Hout"], g["Wout"], g["stride"], g["pad"], g["T"], g["HW"]
    d.slices, d.chunk = slices, chunk
    hip.check(hip.lib().mudg_wgrad(C.byref(d), _s()), "mudg_wgrad")
    return (slabs if slices == 1 else group_colsum(slabs)).reshape(m, n)


def temporal_attention_bwd(qkv, do, clips, t, hw, heads, scale):
    _f32(qkv); _f32(do)
    c = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    esz = 4
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + c * esz, qkv.data_ptr() + 2 * c * esz
    dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + c * esz, dqkv.data_ptr() + 2 * c * esz
    hip.check(hip.lib().mudg_temporal_attention_bwd(q, k, v, do.data_ptr(), qkv.stride(0), do.stride(0), dq, dk, dv, dqkv.stride(0), clips, t, hw,
                                                    heads, scale, _s()), "mudg_temporal_attention_bwd")
    return dqkv


def mse(pred, target, weights=None, want_grad=False):
    """(loss per sample [B], gradient of sum_b w[b] loss[b] or None) for (B, ...) fp32 tensors."""
    pred, target = pred.contiguous(), target.contiguous()
    b = pred.shape[0]
    n = pred.numel() // b
    loss = torch.empty(b, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    ws = torch.empty(hip.lib().mudg_mse_ws_doubles(b), dtype=torch.float64, device=pred.device)
    hip.check(hip.lib().mudg_mse(pred.data_ptr(), target.data_ptr(), None if weights is None else weights.data_ptr(), b, n, loss.data_ptr(),
                                 None if grad is None else grad.data_ptr(), ws.data_ptr(), _s()), "mudg_mse")
    return loss, grad


def upsample2x(x, frames, h, w, adjoint=False):
    _f32(x)
    c = x.shape[1]
    x = x.contiguous()
    out = torch.empty((frames * h * w * (1 if adjoint else 4), c), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().mudg_upsample2x(x.data_ptr(), out.data_ptr(), frames, h, w, c, int(adjoint), _s()), "mudg_upsample2x")
    return out


def dilate2x(dy, frames, ho, wo, hi, wi):
    _f32(dy)
    c = dy.shape[1]
    dy = dy.contiguous()
    out = torch.empty((frames * hi * wi, c), dtype=torch.float32, device=dy.device)
    hip.check(hip.lib().mudg_dilate2x(dy.data_ptr(), out.data_ptr(), frames, ho, wo, hi, wi, c, _s()), "mudg_dilate2x")
    return out


def silu(x, dy=None):
    x = x.contiguous()
    out = torch.empty_like(x)
    hip.check(hip.lib().mudg_silu(x.data_ptr(), None if dy is None else dy.contiguous().data_ptr(), out.data_ptr(), x.numel(), _s()), "mudg_silu")
    return out


def gelu(x, dy=None):
    """Exact (erf) GELU, or dy * gelu'(x) when dy is given."""
    x = x.contiguous()
    out = torch.empty_like(x)
    hip.check(hip.lib().mudg_gelu(x.data_ptr(), None if dy is None else dy.contiguous().data_ptr(), out.data_ptr(), x.numel(), _s()), "mudg_gelu")
    return out


def adamw_multi_(table, nchunks, *, lr, betas, eps, weight_decay, step):
    """One torch.optim.AdamW step over every tensor listed in `table` (device int64 [nchunks][5]: p, g, m, v, count), one launch."""
    hip.check(hip.lib().mudg_adamw_multi(table.data_ptr(), nchunks, lr, betas[0], betas[1], eps, weight_decay, step, _s()), "mudg_adamw_multi")


def adamw_(p, g, m, v, *, lr, betas, eps, weight_decay, step):
    """One torch.optim.AdamW step on flat fp32 buffers, in place."""
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise hip.MudgError("adamw_ expects contiguous fp32 tensors")
    hip.check(hip.lib().mudg_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, betas[0], betas[1], eps, weight_decay,
                                   step, _s()), "mudg_adamw")


def dropout(x, p, seed):
    x = x.contiguous()
    out = torch.empty_like(x)
    hip.check(hip.lib().mudg_dropout(x.data_ptr(), out.data_ptr(), x.numel(), p, seed, _s()), "mudg_dropout")
    return out
