"""Tensor-level wrappers over the C-ABI: torch tensors are used purely as device buffers (pointer, stride,
stream); every arithmetic result on the hot path comes out of a HIP kernel in libmudg_hip.so.

Activations are "rows" matrices: 2-D bf16 tensors [pixels, channels] whose row stride may exceed the channel
count (views into wider buffers are fine as long as stride(1) == 1 and rows are 16-byte aligned).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import hip



def H16():
    """torch dtype of the 16-bit MFMA operands of the loaded library (bf16 by default, fp16 with MUDG_OPERAND=fp16)."""
    return hip.operand_dtype()


def STREAM():
    """torch dtype of the residual stream — the tensors later layers add onto (block outputs, the transformers' token
    stream, encoder skips): fp16 with bf16 operands (2 bytes per value through HBM; its 11 significand bits keep the ~150
    residual adds below the bf16 operand rounding, and the reference's own stream is fp16 under torch.autocast), fp32 in
    the accuracy-oriented modes — fp16 operands (an fp16 stream measured +35 % error per forward there: 2.0e-3 -> 2.7e-3)
    and the split-operand precision builds.  MUDG_STREAM=fp32 forces fp32."""
    import os
    if hip.operand_name() != "bf16" or os.environ.get("MUDG_STREAM", "").lower() == "fp32":
        return torch.float32
    return torch.float16


def kind(t) -> int:
    """Storage code of a rows matrix for the C-ABI (out_fp32 / res_fp32 / x_fp32 ...): 0 operand, 1 fp32, 2 fp16."""
    if t.dtype == torch.float32:
        return 1
    if t.dtype == H16():
        return 0
    if t.dtype == torch.float16:
        return 2
    raise hip.MudgError(f"no storage code for {t.dtype}")


def _out_dtype(out_fp32, out_stream):
    return torch.float32 if out_fp32 else (STREAM() if out_stream else H16())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rows(t: torch.Tensor, dtype=None) -> torch.Tensor:
    dtype = dtype or H16()
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != dtype or not t.is_cuda:
        raise hip.MudgError(f"expected a cuda {dtype} rows matrix with unit channel stride, got "
                            f"{tuple(t.shape)} {t.dtype} strides {t.stride()} on {t.device}")
    return t


def empty_rows(rows: int, cols: int, dtype=None, device=None) -> torch.Tensor:
    """A rows matrix.  Operand matrices of the split-operand builds (hip.planes() > 1) are allocated planes * cols wide
    and returned as the [rows, cols] view of piece 0: piece p of a row starts stride(0) / planes elements further, which
    every column slice of the view inherits (the kernels derive the plane distance from the row stride)."""
    dtype = dtype or H16()
    planes = hip.planes() if dtype == H16() else 1
    if planes > 1:
        return torch.empty((rows, planes * cols), dtype=dtype, device=device or "cuda")[:, :cols]
    return torch.empty((rows, cols), dtype=dtype, device=device or "cuda")


def repeat_rows(t, times):
    """`times` copies of a rows matrix stacked along the rows (batch-major rows: the batch repeated).  GroupNorm partial
    sums hanging on `t` (one set per 128-row block) are repeated with it when the blocks line up."""
    if times == 1:
        return t
    rows, cols = t.shape
    out = empty_rows(times * rows, cols, t.dtype, t.device)
    src, dst = t, out
    if out._base is not None:            # an operand matrix of a split-operand build: every piece is copied
        if t._base is None or tuple(t._base.shape) != (rows, out._base.shape[1]):
            raise hip.MudgError("repeat_rows: not a whole operand matrix")
        src, dst = t._base, out._base
    for i in range(times):
        dst[i * rows:(i + 1) * rows].copy_(src)
    ws = getattr(t, GN_ATTR, None)
    brows = getattr(t, GN_ATTR + "_rows", 128)
    if ws is not None and getattr(t, GN_ATTR + "_version", -1) == _version(t) and rows % brows == 0 and not getattr(t, GN_ATTR + "_clips", 0):
        setattr(out, GN_ATTR, ws.repeat(times, 1, 1))
        setattr(out, GN_ATTR + "_version", _version(out))
        setattr(out, GN_ATTR + "_rows", brows)
    return out


# ------------------------------------------------------------------------------------------------ GEMM family
GN_ATTR = "_mudg_gn_partials"      # python attribute a producer leaves on its output: fp32 [ceil(M/rows)][N][2] partial sums per block of
                                   # `rows` rows (attribute GN_ATTR + "_rows": 128, or 288 where the 288 x 320-tile kernel ran the problem)


def _drop_stats(t):
    """A kernel is about to write into `t` through its raw pointer (torch's version counter does not see that): partial
    sums a previous producer hung on it would be stale."""
    if t is not None and getattr(t, GN_ATTR, None) is not None:
        setattr(t, GN_ATTR, None)
        setattr(t, GN_ATTR + "_clips", 0)


def _version(t):
    try:
        return t._version
    except RuntimeError:            # inference tensors do not track versions
        return 0


def _attach_stats(d, out, M, nout):
    """Ask the epilogue for GroupNorm partials of `out` (MudgGemmDesc.stats) and hang them on the tensor.  Call it on the finished
    descriptor: the height of the partial blocks depends on which kernel the library will run (mudg_gemm_stats_rows)."""
    brows = hip.lib().mudg_gemm_stats_rows(C.byref(d))
    ws = torch.empty(((M + brows - 1) // brows, nout, 2), dtype=torch.float32, device=out.device)
    d.stats = ws.data_ptr()
    setattr(out, GN_ATTR + "_rows", brows)
    setattr(out, GN_ATTR, ws)
    setattr(out, GN_ATTR + "_version", _version(out))     # a later torch in-place op on `out` invalidates the partials
    setattr(out, GN_ATTR + "_clips", 0)                   # (the tiling tag of an earlier slab-major tconv3 into the same `out=` is stale)


def gemm(x, w, *, out=None, bias=None, gbias=None, rows_per_group=0, residual=None, x2=None, geglu=False,
         out_fp32=False, alpha=1.0, batch=1, sx=0, sw=0, sy=0, sr=0, M=None, N=None, K=None, ldy=None, gelu=False,
         stats=False, out_stream=False, fp8=False, frame_rows=0):
    """out[m, n] = epilogue(alpha * sum_k x[m, k] w[n, k]); see MudgGemmDesc.  frame_rows: a hint — the rows of one frame of x
    (results never depend on it; it lets the library pick a tile height that divides a frame).  The result is an MFMA operand matrix by
    default, fp32 with out_fp32, the residual-stream dtype (STREAM()) with out_stream; `out=` decides by its dtype.
    fp8=True (16-bit builds, operand result, N % 32 == 0): returns (out, e4m3 bytes [M, N] uint8, E8M0 scales [M, N / 32] uint8) —
    the MX-fp8 copy of the result written by the same epilogue (MudgGemmDesc.Y8), bit-equal to quantize_mxfp8(out)."""
    _rows(x); _rows(w)
    if M is None:
        M = x.shape[0]
    if N is None:
        N = w.shape[0]
    if K is None:
        K = w.shape[1]
    nout = N // 2 if geglu else N
    if out is None:
        out = empty_rows(M, nout, _out_dtype(out_fp32, out_stream), x.device)
    else:
        _drop_stats(out)
    d = hip.GemmDesc()
    d.X, d.X2, d.W, d.Y = x.data_ptr(), _ptr(x2), w.data_ptr(), out.data_ptr()
    d.bias, d.gbias, d.R = _ptr(bias), _ptr(gbias), _ptr(residual)
    d.M, d.N, d.K = M, N, K
    d.ldx, d.ldw = x.stride(0), w.stride(0)
    d.ldx2 = x2.stride(0) if x2 is not None else 0
    d.ldy = ldy if ldy is not None else out.stride(0)
    d.ldr = residual.stride(0) if residual is not None else 0
    d.csplit = x.shape[1] if x2 is not None else K
    d.batch, d.sX, d.sW, d.sY, d.sR = batch, sx, sw, sy, sr
    d.rows_per_group, d.out_fp32, d.geglu, d.alpha, d.mode = rows_per_group, kind(out), int(geglu), alpha, 0
    d.res_fp32 = kind(residual) if residual is not None else 0
    d.act = int(gelu)
    d.HW = int(frame_rows)
    if fp8:
        y8 = torch.empty((M, nout), dtype=torch.uint8, device=x.device)
        s8 = torch.empty((M, nout // 32), dtype=torch.uint8, device=x.device)
        d.Y8, d.S8, d.ldy8, d.lds8 = y8.data_ptr(), s8.data_ptr(), y8.stride(0), s8.stride(0)
    if stats:
        _attach_stats(d, out, M, nout)
    hip.check(hip.lib().mudg_gemm(C.byref(d), _stream()), "mudg_gemm")
    return (out, y8, s8) if fp8 else out


def conv3x3(x, w, *, frames, hin, win, cin, stride=1, upsample=False, out=None, bias=None, gbias=None,
            rows_per_group=0, residual=None, x2=None, out_fp32=False, korder=0, pad=1, stats=False, out_stream=False):
    """3x3 / pad 1 convolution on channels-last rows; w is packed [Cout][9*cin], K axis tap-major (korder 0) or
    64-channel-slab-major (korder 1, see MudgGemmDesc.korder)."""
    _rows(x); _rows(w)
    if upsample:
        hout, wout = 2 * hin, 2 * win
    elif pad == 1:
        hout, wout = (hin - 1) // stride + 1, (win - 1) // stride + 1
    else:       # pad 0 with one trailing zero row / column: AutoencoderKL's downsample
        hout, wout = (hin + 1 - 3) // stride + 1, (win + 1 - 3) // stride + 1
    M, N = frames * hout * wout, w.shape[0]
    if out is None:
        out = empty_rows(M, N, _out_dtype(out_fp32, out_stream), x.device)
    else:
        _drop_stats(out)
    d = hip.GemmDesc()
    d.X, d.X2, d.W, d.Y = x.data_ptr(), _ptr(x2), w.data_ptr(), out.data_ptr()
    d.bias, d.gbias, d.R = _ptr(bias), _ptr(gbias), _ptr(residual)
    d.out_fp32 = kind(out)
    d.res_fp32 = kind(residual) if residual is not None else 0
    d.M, d.N, d.K = M, N, 9 * cin
    d.ldx, d.ldw, d.ldy = x.stride(0), w.stride(0), out.stride(0)
    d.ldx2 = x2.stride(0) if x2 is not None else 0
    d.ldr = residual.stride(0) if residual is not None else 0
    d.csplit = x.shape[1] if x2 is not None else cin
    d.batch, d.rows_per_group, d.alpha, d.mode = 1, rows_per_group, 1.0, 1
    d.Hin, d.Win, d.Hout, d.Wout, d.Cin, d.stride, d.upsample = hin, win, hout, wout, cin, stride, int(upsample)
    d.korder, d.pad = korder, pad
    if stats:
        _attach_stats(d, out, M, N)
    hip.check(hip.lib().mudg_gemm(C.byref(d), _stream()), "mudg_gemm[conv3x3]")
    return out


def conv3x3_up2(x, wsub, *, frames, hin, win, cin, bias=None, out_fp32=False, out_stream=False):
    """Nearest-2x upsample followed by a 3x3 / pad 1 conv in the sub-pixel form — four 2x2 convs on the low-resolution
    image, 4/9 of the multiply-adds (MudgGemmDesc.subpixel; `wsub` from packing.conv3x3_subpixel).  Returns None when the
    problem is outside what the descriptor loader accepts: the caller then runs conv3x3(upsample=True) with the 3x3 weights."""
    _rows(x); _rows(wsub)
    N = wsub.shape[0] // 4
    M = frames * hin * win
    d = hip.GemmDesc()
    d.X, d.W, d.bias = x.data_ptr(), wsub.data_ptr(), _ptr(bias)
    d.M, d.N, d.K = M, N, 4 * cin
    d.ldx, d.ldw = x.stride(0), wsub.stride(0)
    d.csplit, d.batch, d.alpha, d.mode = cin, 4, 1.0, 1
    d.sW = N * wsub.stride(0)
    d.Hin, d.Win, d.Hout, d.Wout, d.Cin, d.stride, d.upsample = hin, win, hin, win, cin, 1, 0
    d.korder, d.pad, d.subpixel = 1, 1, 1
    if not hip.lib().mudg_conv_subpixel_ok(C.byref(d)):
        return None
    out = empty_rows(4 * M, N, _out_dtype(out_fp32, out_stream), x.device)
    d.Y, d.ldy, d.out_fp32 = out.data_ptr(), out.stride(0), kind(out)
    hip.check(hip.lib().mudg_gemm(C.byref(d), _stream()), "mudg_gemm[conv3x3 subpixel]")
    return out


def tconv3_slab_ok(t, hw, cin):
    """Whether mudg_gemm accepts korder = 1 for this temporal conv (MudgGemmDesc.korder, mode 2)."""
    return hip.planes() <= 2 and t == 16 and hw % 8 == 0 and cin % 64 == 0


def tconv3_wide(t, hw, cin, cout):
    """Whether the library runs this temporal conv (plain K order, korder 0) on a tile kernel of wgemm.hip (288 x 320 or 160 x 320) — the caller then does not
    ask for the slab-major order, whose 8-pixel x 16-frame tiles belong to the 128 x 128 kernels.  A dry query: nothing is read.
    Not cached: it is one host call, and the variant builds re-read MUDG_GEMM_W288 at every call (a cached answer went stale when a
    test toggled the switch in-process).  The query assumes what the executor always passes — dense, 16-byte-aligned rows; a caller
    with other strides gets a correct result either way (ops.tconv3 asks again with the real descriptor for the GroupNorm block
    height), only possibly the slower of the two K orders."""
    d = hip.GemmDesc()
    d.X, d.W, d.Y = 256, 256, 256                     # aligned placeholders
    d.M, d.N, d.K = 16 * t * hw, cout, 3 * cin
    pl = hip.planes()
    d.ldx, d.ldw, d.ldy = pl * cin, pl * 3 * cin, pl * cout
    d.csplit, d.batch, d.alpha, d.mode = cin, 1, 1.0, 2
    d.Cin, d.T, d.HW = cin, t, hw
    return hip.lib().mudg_gemm_stats_rows(C.byref(d)) != 128       # 288 or 160: a tile kernel of wgemm.hip


def tconv3(x, w, *, clips, t, hw, cin, out=None, bias=None, residual=None, out_fp32=False, stats=False, out_stream=False, korder=0):
    """(3,1,1) temporal convolution, pad (1,0,0), on rows ordered ((b t) hw); w packed [Cout][3*cin], K axis [tap][cin] (korder 0)
    or [cin/64][tap][64] (korder 1: 16-frame clips; the kernels then tile a clip as 8 pixels x 16 frames and stage a 64-channel slab
    once for the three taps).  With korder = 1 the GroupNorm partials of `stats` are per such TILE, not per 128 consecutive rows:
    they are tagged and only a clip-level GroupNorm (samples = clips) takes them."""
    _rows(x); _rows(w)
    M, N = clips * t * hw, w.shape[0]
    if out is None:
        out = empty_rows(M, N, _out_dtype(out_fp32, out_stream), x.device)
    else:
        _drop_stats(out)
    d = hip.GemmDesc()
    d.X, d.W, d.Y = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias, d.R = _ptr(bias), _ptr(residual)
    d.out_fp32 = kind(out)
    d.res_fp32 = kind(residual) if residual is not None else 0
    d.M, d.N, d.K = M, N, 3 * cin
    d.ldx, d.ldw, d.ldy = x.stride(0), w.stride(0), out.stride(0)
    d.ldr = residual.stride(0) if residual is not None else 0
    d.csplit, d.batch, d.alpha, d.mode = cin, 1, 1.0, 2
    d.Cin, d.T, d.HW = cin, t, hw
    d.korder = korder
    if stats:
        _attach_stats(d, out, M, N)
        setattr(out, GN_ATTR + "_clips", clips if korder else 0)
    hip.check(hip.lib().mudg_gemm(C.byref(d), _stream()), "mudg_gemm[tconv3]")
    return out


# ------------------------------------------------------------------------------------------------ attention
def attention(q, k, vt, out, *, frames, heads, nq, nk, ldvt=None, svt=None, kv_div=1, scale=0.125, accumulate=False,
              k2=None, vt2=None, nk2=0, ldvt2=None, svt2=None, kv_div2=1, q_prescaled=False, fp8=None, lse=None):
    """vt: V^T as [kv batches * heads * 64, keys] rows (row stride = ldvt, batch stride = heads * 64 rows by default).
    k2 / vt2 / nk2: an optional second key / value set with its own softmax whose output is added (the image tokens of
    the text + image cross-attention), in the same launch.  q_prescaled: q already carries scale * log2(e) (folded into
    the packed q-projection weights); `scale` is then ignored and long self-attention runs its lean softmax.
    fp8 = (q8, qs, k8, ks) from quantize_mxfp8: Q K^T on the MX-fp8 MFMA (long self-attention, needs q_prescaled)."""
    if ldvt is None:
        ldvt = vt.stride(0)
    if svt is None:
        svt = heads * 64 * ldvt
    _drop_stats(out)
    d = hip.AttnDesc()
    d.Q, d.K, d.Vt, d.O = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    d.F, d.heads, d.Nq, d.Nk = frames, heads, nq, nk
    d.ldq, d.ldk, d.ldvt, d.ldo = q.stride(0), k.stride(0), ldvt, out.stride(0)
    d.svt, d.kv_div, d.scale, d.accumulate = svt, kv_div, scale, int(accumulate)
    d.q_prescaled = int(q_prescaled)
    if lse is not None:              # fp32 [frames * nq][heads]: the softmax statistics the training backward pass reuses
        if lse.dtype != torch.float32 or tuple(lse.shape) != (frames * nq, heads) or not lse.is_contiguous():
            raise hip.MudgError("attention: lse must be a contiguous fp32 [frames * nq][heads] tensor")
        d.Lse = lse.data_ptr()
    if fp8 is not None:
        q8, qs, k8, ks = fp8
        d.Q8, d.Qs, d.K8, d.Ks = q8.data_ptr(), qs.data_ptr(), k8.data_ptr(), ks.data_ptr()
        d.ldq8, d.ldqs, d.ldk8, d.ldks = q8.stride(0), qs.stride(0), k8.stride(0), ks.stride(0)
    if k2 is not None:
        d.K2, d.Vt2, d.Nk2, d.ldk2 = k2.data_ptr(), vt2.data_ptr(), nk2, k2.stride(0)
        d.ldvt2 = vt2.stride(0) if ldvt2 is None else ldvt2
        d.svt2 = heads * 64 * d.ldvt2 if svt2 is None else svt2
        d.kv_div2 = kv_div2
    hip.check(hip.lib().mudg_attention(C.byref(d), _stream()), "mudg_attention")
    return out


def quantize_mxfp8(x):
    """Operand rows [rows, cols] (cols % 32 == 0) -> (e4m3 bytes [rows, cols] uint8, E8M0 scales [rows, cols / 32] uint8):
    OCP microscaling, one power-of-two scale per 32 consecutive columns (mudg_quantize_mxfp8)."""
    _rows(x)
    rows, cols = x.shape
    y = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows, cols // 32), dtype=torch.uint8, device=x.device)
    hip.check(hip.lib().mudg_quantize_mxfp8(x.data_ptr(), x.stride(0), rows, cols, y.data_ptr(), y.stride(0), sc.data_ptr(),
                                            sc.stride(0), _stream()), "mudg_quantize_mxfp8")
    return y, sc


def temporal_attention(qkv, out, *, clips, t, hw, heads, scale=0.125):
    hip.check(hip.lib().mudg_temporal_attention(qkv.data_ptr(), out.data_ptr(), clips, t, hw, heads,
                                                qkv.stride(0), out.stride(0), scale, _stream()),
              "mudg_temporal_attention")
    return out


# ------------------------------------------------------------------------------------------------ norms
def _rows_any(t):
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype not in (H16(), torch.float32, torch.float16) or not t.is_cuda:
        raise hip.MudgError(f"expected a cuda operand / fp32 / fp16 rows matrix, got {tuple(t.shape)} {t.dtype} on {t.device}")
    return t


def groupnorm(x, gamma, beta, *, samples, rows, eps, silu, groups=32, x2=None, out=None, fused=True, return_stats=False):
    """GroupNorm(+SiLU).  When every source tensor still carries the partial sums its producing GEMM / conv wrote
    (stats=True there) and a sample is a whole number of 128-row blocks, the statistics pass over x is skipped.
    return_stats: also return the (mean, rstd) per (sample, group) the kernels computed, fp32 [samples * groups][2] (the
    backward pass of the training step needs them)."""
    _rows_any(x)
    if x2 is not None and x2.dtype != x.dtype:
        raise hip.MudgError("groupnorm: both channel sources must share a dtype")
    c = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
    if out is None:
        out = empty_rows(samples * rows, c, H16(), x.device)
    def partials(t):
        ws = getattr(t, GN_ATTR, None)
        tiled = getattr(t, GN_ATTR + "_clips", 0)          # partials per (8 pixels x 16 frames) tile of a clip (tconv3, korder 1)
        if tiled and tiled != samples:
            return None, 128
        ok_ = ws is not None and getattr(t, GN_ATTR + "_version", -1) == _version(t)
        return (ws if ok_ else None), getattr(t, GN_ATTR + "_rows", 128)

    p1, r1 = partials(x) if fused else (None, 128)
    p2, r2 = partials(x2) if (fused and x2 is not None) else (None, r1)
    ok = p1 is not None and rows % r1 == 0 and p1.shape[1] == x.shape[1] and p1.shape[0] * r1 >= samples * rows
    if ok and x2 is not None:
        ok = p2 is not None and rows % r2 == 0 and p2.shape[1] == x2.shape[1] and p2.shape[0] * r2 >= samples * rows
    if ok:
        ws = torch.empty(2 * samples * groups, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().mudg_groupnorm_fused_rows(x.data_ptr(), _ptr(x2), x.shape[1], x.stride(0),
                                                      x2.stride(0) if x2 is not None else 0, kind(x),
                                                      gamma.data_ptr(), beta.data_ptr(),
                                                      out.data_ptr(), out.stride(0), samples, rows, c, groups, eps, int(silu),
                                                      p1.data_ptr(), r1, _ptr(p2), r2, ws.data_ptr(), _stream()), "mudg_groupnorm_fused")
        return (out, ws.reshape(samples * groups, 2)) if return_stats else out
    n = hip.lib().mudg_groupnorm_ws_floats(samples, groups, rows)
    ws = torch.empty(n, dtype=torch.float32, device=x.device)
    hip.check(hip.lib().mudg_groupnorm(x.data_ptr(), _ptr(x2), x.shape[1], x.stride(0),
                                       x2.stride(0) if x2 is not None else 0, kind(x),
                                       gamma.data_ptr(), beta.data_ptr(),
                                       out.data_ptr(), out.stride(0), samples, rows, c, groups, eps, int(silu),
                                       ws.data_ptr(), _stream()), "mudg_groupnorm")
    # the statistics sit at the tail of the scratch (behind the chunk partials): see mudg_groupnorm
    return (out, ws[-2 * samples * groups:].reshape(samples * groups, 2)) if return_stats else out


def layernorm(x, gamma, beta, *, eps=1e-5, out=None):
    _rows_any(x)
    if out is None:
        out = empty_rows(x.shape[0], x.shape[1], H16(), x.device)
    hip.check(hip.lib().mudg_layernorm(x.data_ptr(), x.stride(0), kind(x), gamma.data_ptr(),
                                       beta.data_ptr(), out.data_ptr(),
                                       out.stride(0), x.shape[0], x.shape[1], eps, _stream()), "mudg_layernorm")
    return out


def softmax_rows(s, out=None):
    if out is None:
        out = empty_rows(s.shape[0], s.shape[1], H16(), s.device)
    hip.check(hip.lib().mudg_softmax_rows(s.data_ptr(), s.stride(0), out.data_ptr(), out.stride(0), s.shape[0],
                                          s.shape[1], _stream()), "mudg_softmax_rows")
    return out


# ------------------------------------------------------------------------------------------------ small stuff
_FREQS = {}


def sinusoid_freqs(dim, max_period, device):
    """exp(-ln(max_period) * arange(dim/2) / (dim/2)) in fp32 on the HOST, op for op as utils_diffusion.py:19-22."""
    key = (dim, float(max_period), str(device))
    if key not in _FREQS:
        half = dim // 2
        f = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
        _FREQS[key] = f.to(device)
    return _FREQS[key]


def timestep_embedding(t, dim, max_period=10000):
    t = t.to(torch.int64).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    freqs = sinusoid_freqs(dim, max_period, t.device)
    hip.check(hip.lib().mudg_timestep_embedding(t.data_ptr(), freqs.data_ptr(), out.data_ptr(), t.shape[0], dim,
                                                _stream()), "mudg_timestep_embedding")
    return out


def small_linear(x, w, b=None, *, act_in=False, act_out=False, out=None, accumulate=False):
    """fp32 x [M, K] times w [N, K] (fp32 or bf16) plus bias, optional SiLU before / after."""
    if x.dtype != torch.float32 or not x.is_contiguous() or not w.is_contiguous():
        raise hip.MudgError("small_linear expects contiguous fp32 x and contiguous w")
    m, k = x.shape
    n = w.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().mudg_small_linear(x.data_ptr(), w.data_ptr(), int(w.dtype == H16()), _ptr(b), out.data_ptr(),
                                          m, n, k, int(act_in), int(act_out), int(accumulate), _stream()),
              "mudg_small_linear")
    return out


def ncthw_to_rows(src, dst, coff=0, t0=0, frames=None):
    """(B, C, T, H, W) fp32|bf16 -> dst rows ((b t) h w) channels [coff, coff + C); optional frame window
    [t0, t0 + frames) of the T axis."""
    b, c, t, h, w = src.shape
    src = src.contiguous()
    n = t if frames is None else frames
    hip.check(hip.lib().mudg_ncthw_to_rows(src.data_ptr(), int(src.dtype == torch.float32), dst.data_ptr(), b, c, n,
                                           h * w, dst.stride(0), coff, t, t0, _stream()), "mudg_ncthw_to_rows")
    return dst


def rows_to_ncthw(src, shape, coff=0, dtype=torch.float32, scale=1.0, out=None, t0=0, frames=None):
    """rows ((b t) h w) -> (B, C, T, H, W); with `out` given, writes frames [t0, t0 + frames) of an existing tensor."""
    b, c, t, h, w = shape
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=src.device)
    elif tuple(out.shape) != tuple(shape) or not out.is_contiguous():
        raise hip.MudgError("rows_to_ncthw: `out` must be a contiguous tensor of the stated shape")
    n = t if frames is None else frames
    hip.check(hip.lib().mudg_rows_to_ncthw(src.data_ptr(), kind(src), src.stride(0), coff,
                                           out.data_ptr(), int(out.dtype == torch.float32), b, c, n, h * w, scale,
                                           t, t0, _stream()),
              "mudg_rows_to_ncthw")
    return out


def zero_channels(dst, c0, c1):
    hip.check(hip.lib().mudg_zero_channels(dst.data_ptr(), dst.shape[0], dst.stride(0), c0, c1, _stream()),
              "mudg_zero_channels")
    return dst


def cast_rows(src, dst):
    """dst[r, c] = src[r, c] between rows matrices of either kind (operand or fp32), any direction (mudg_cast_rows)."""
    if src.dim() != 2 or dst.dim() != 2 or src.shape != dst.shape or src.stride(1) != 1 or dst.stride(1) != 1:
        raise hip.MudgError(f"cast_rows: expected equal-shape rows matrices, got {tuple(src.shape)} -> {tuple(dst.shape)}")
    for t in (src, dst):
        if not t.is_cuda:
            raise hip.MudgError(f"cast_rows: expected cuda tensors, got {t.device}")
    hip.check(hip.lib().mudg_cast_rows(src.data_ptr(), kind(src), src.stride(0), dst.data_ptr(),
                                       kind(dst), dst.stride(0), src.shape[0], src.shape[1], _stream()),
              "mudg_cast_rows")
    return dst


def cast_bf16(src):
    """fp32 / stream rows matrix -> MFMA operand rows of the same shape (an operand matrix is returned as is)."""
    if src.dtype == H16():
        return src
    if src.dtype == torch.float16 and src.dim() == 2 and src.stride(1) == 1:        # fp16 stream -> bf16 operand
        return cast_rows(src, empty_rows(src.shape[0], src.shape[1], H16(), src.device))
    if src.dtype != torch.float32:
        raise hip.MudgError("cast_bf16 expects fp32 (or a 2-D stream matrix)")
    if src.dim() == 2 and src.stride(1) == 1:
        return cast_rows(src, empty_rows(src.shape[0], src.shape[1], H16(), src.device))
    if hip.planes() > 1 or not src.is_contiguous():
        raise hip.MudgError("cast_bf16: only 2-D rows matrices have an operand layout in the split-operand builds")
    out = torch.empty(src.shape, dtype=H16(), device=src.device)         # 16-bit builds: any contiguous shape, flat
    hip.check(hip.lib().mudg_cast_f32_bf16(src.data_ptr(), out.data_ptr(), src.numel(), _stream()), "mudg_cast_f32_bf16")
    return out


def to_f32(src):
    """Operand rows matrix -> fp32 copy of the same shape."""
    if src.dim() != 2:
        raise hip.MudgError("to_f32 expects a 2-D rows matrix")
    return cast_rows(src, torch.empty(src.shape, dtype=torch.float32, device=src.device))


def copy_rows(src, dst):
    """dst[r, :] = src[r, :] for bf16 2-D views with unit inner stride."""
    _rows(src); _rows(dst)
    hip.check(hip.lib().mudg_copy_rows(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), src.shape[0],
                                       src.shape[1], _stream()), "mudg_copy_rows")
    return dst


def add_(y, x, alpha=1.0):
    """y += alpha * x on contiguous fp32 tensors of equal size."""
    if y.dtype != torch.float32 or x.dtype != torch.float32 or not (y.is_contiguous() and x.is_contiguous()):
        raise hip.MudgError("add_ expects contiguous fp32 tensors")
    _drop_stats(y)
    hip.check(hip.lib().mudg_axpy_f32(y.data_ptr(), x.data_ptr(), y.numel(), alpha, _stream()), "mudg_axpy_f32")
    return y


def lincomb(x, y, ca, cb):
    """ca[b] * x[b] + cb[b] * y[b] for fp32 (B, ...) tensors and fp32 [B] device coefficient vectors."""
    x, y = x.float().contiguous(), y.float().contiguous()
    ca, cb = ca.float().contiguous(), cb.float().contiguous()
    out = torch.empty_like(x)
    b = x.shape[0]
    hip.check(hip.lib().mudg_lincomb(out.data_ptr(), x.data_ptr(), y.data_ptr(), ca.data_ptr(), cb.data_ptr(), b,
                                     x.numel() // b, _stream()), "mudg_lincomb")
    return out


def gaussian_sample(moments, noise=None, scale=1.0):
    """moments (N, 2C, H, W) fp32 -> scale * (mean + std * noise) as (N, C, H, W); noise None = mode."""
    moments = moments.float().contiguous()
    n, c2, h, w = moments.shape
    out = torch.empty((n, c2 // 2, h, w), dtype=torch.float32, device=moments.device)
    if noise is not None:
        noise = noise.to(device=moments.device, dtype=torch.float32).contiguous()
    hip.check(hip.lib().mudg_gaussian_sample(moments.data_ptr(), _ptr(noise), out.data_ptr(), n, c2 // 2, h * w, scale,
                                             _stream()), "mudg_gaussian_sample")
    return out


def ddim_step(x, e_c, e_u, noise, coef, e_m=None):
    """Fused DDIM update on fp32 latents (B, ...). coef = 8 to 10 host floats, see mudg_ddim_step; e_m = the
    image-only-conditioned pass of the three-way guidance (coef[8] = cfg_img); coef[9] = 1 for eps-predicting models."""
    def dense(t):      # raw pointers are handed to the kernel: insist on dense fp32 (permuted views are copied)
        return None if t is None else t.to(torch.float32).contiguous()

    x, e_c, e_u, noise, e_m = dense(x), dense(e_c), dense(e_u), dense(noise), dense(e_m)
    coef = list(coef) + [0.0] * (10 - len(coef))
    b = x.shape[0]
    n = x.numel() // b
    x_prev, pred_x0 = torch.empty_like(x), torch.empty_like(x)
    ws = torch.empty(hip.lib().mudg_ddim_ws_doubles(b), dtype=torch.float64, device=x.device)
    arr = (C.c_float * 10)(*[float(v) for v in coef])
    hip.check(hip.lib().mudg_ddim_step(x.data_ptr(), e_c.data_ptr(), _ptr(e_u), _ptr(e_m), _ptr(noise), x_prev.data_ptr(),
                                       pred_x0.data_ptr(), b, n, arr, ws.data_ptr(), _stream()), "mudg_ddim_step")
    return x_prev, pred_x0


# ------------------------------------------------------------------------------------------------ post-processing
def frames_to_uint8(video):
    """(b, c, t, h, w) float -> (b, t, h, w, c) uint8: clamp to [-1, 1], (x + 1) / 2 * 255, truncate (eval_tools.py:22-27)."""
    if video.dim() != 5 or not video.is_cuda:
        raise hip.MudgError(f"frames_to_uint8: expected a cuda (b, c, t, h, w) tensor, got {tuple(video.shape)} on {video.device}")
    v = video.to(torch.float32).contiguous()
    b, c, t, h, w = v.shape
    out = torch.empty((b, t, h, w, c), dtype=torch.uint8, device=v.device)
    hip.check(hip.lib().mudg_frames_to_u8(v.data_ptr(), out.data_ptr(), b, c, t, h * w, _stream()), "mudg_frames_to_u8")
    return out


def depth_from_uint8(frames):
    """(..., h, w, 3) uint8 -> (..., 1, h, w) float32 in [0, 1]: channel mean / 255 (eval_tools.py:71)."""
    if frames.dtype != torch.uint8 or frames.shape[-1] != 3 or not frames.is_cuda:
        raise hip.MudgError("depth_from_uint8: expected a cuda (..., h, w, 3) uint8 tensor")
    f = frames.contiguous()
    out = torch.empty(f.shape[:-3] + (1,) + f.shape[-3:-1], dtype=torch.float32, device=f.device)
    hip.check(hip.lib().mudg_depth_from_u8(f.data_ptr(), out.data_ptr(), f.numel() // 3, _stream()), "mudg_depth_from_u8")
    return out


def semantic_nearest(img):
    """(3, h, w) uint8 -> ((3, h, w) uint8 recoloured, (h, w) int64 labels): nearest of 19 palette colours
    (eval_tools.py:309-347)."""
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[0] != 3 or not img.is_cuda:
        raise hip.MudgError("semantic_nearest: expected a cuda (3, h, w) uint8 tensor")
    x = img.contiguous()
    vis = torch.empty_like(x)
    lab = torch.empty(x.shape[1:], dtype=torch.int64, device=x.device)
    hip.check(hip.lib().mudg_semantic_nearest(x.data_ptr(), vis.data_ptr(), lab.data_ptr(), x.shape[1] * x.shape[2], _stream()),
              "mudg_semantic_nearest")
    return vis, lab
