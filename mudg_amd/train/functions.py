"""autograd Functions of the training step: every forward and every backward is a sequence of HIP kernel launches.

Activations and gradients travel between Functions as fp32 rows matrices ((b t) h w) x channels; each Function casts what its
MFMA kernels consume to operand matrices (bf16, or the bf16 pieces of the precision builds) on the way in.  How the backward
contractions map onto the forward GEMM kernel (mudg_gemm: Y[m][n] = sum_k X[m][k] W[n][k]):

    linear      dX = dY W                 X = dY,            W = W^T            (transpose_gather of the weight)
                dW = dY^T X               X = dY^T,          W = X^T            (contraction over the rows)
    conv 3x3    dX = conv(dY, flip(W))    the same implicit-GEMM kernel, filter rotated by 180 degrees, channels swapped;
                                          stride 2: dY is first laid onto the input grid with zeros in between (dilate2x)
                dW[tap] = dY^T X_tap      X_tap^T = transpose_gather(mode 1): the input pixels tap (dy, dx) reads, transposed
    temporal    the same with the three temporal taps (mode 2)
    attention   recomputed: S = scale Q K^T, P = softmax(S), dP = dO V^T, dS = P (dP - rowsum(dP P)) scale,
                dQ = dS K, dK = dS^T Q, dV = P^T dO.  16-bit builds: mudg_attention_bwd (csrc/attention_bwd.hip), the scores
                never leave registers; split-operand builds: per head, fp32 scores in memory, batched GEMMs

Reference semantics: lvdm/modules/attention.py, lvdm/modules/networks/openaimodel3d.py (forward), torch.autograd (backward);
checked against autograd of the CPU oracle in tests/test_training_gpu.py."""
import collections
import weakref

import torch

from .. import hip, ops
from . import kernels as K


_recent_operands = collections.deque(maxlen=4)


def op(t):
    """fp32 rows (any row stride) -> MFMA operand rows.  A producer that computed the operand form anyway (the norms) leaves a
    weak reference to it on its fp32 output; it is taken while it is still alive and the tensor unchanged."""
    tag = getattr(t, "_mudg_operand", None)
    if tag is not None and tag[0] == t._version:
        cached = tag[1]()
        if cached is not None:
            return cached
    return ops.cast_bf16(t)


def _with_operand(y32, y16):
    """Attach the operand form a kernel produced to its fp32 copy; the last few stay alive (the consumers of a norm follow it
    directly), older ones are freed — nothing is added to what the backward pass keeps."""
    _recent_operands.append(y16)
    y32._mudg_operand = (y32._version, weakref.ref(y16))
    return y32


def _pad8(n):
    return (n + 7) // 8 * 8


def _pad_cols(t, cols):
    """Zero-pad the channel axis of fp32 rows to `cols` (layout only)."""
    if t.shape[1] == cols:
        return t
    out = torch.zeros((t.shape[0], cols), dtype=t.dtype, device=t.device)
    out[:, :t.shape[1]].copy_(t)
    return out


def _need(ctx, i):
    return ctx.needs_input_grad[i]


def _splits(m, n, k):
    """K-slices for a weight-gradient GEMM: its output is only (channels x channels) — 9 to 100 tiles for 256 CUs — while the
    contraction runs over every pixel, so the K axis is cut into slices that run as batch entries of the same launch; the
    slices' fp32 results are then summed in a fixed order (group_colsum).  A function of the problem shape only."""
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    want = max(1, min(64, 1024 // max(tiles, 1)))
    return max(1, min(want, k // 512))


def wgrad_gemm(at, bt, m, n, p):
    """sum_p at[m][p] * bt[n][p] -> fp32 [m][n], with the contraction cut into K-slices (see _splits).  at / bt: operand matrices
    [m][ld], [n][ld] whose columns beyond p are zero and whose ld covers slices * chunk."""
    s = _splits(m, n, p)
    if s == 1:
        return ops.gemm(at, bt, out_fp32=True, M=m, N=n, K=at.shape[1])
    chunk = at.shape[1] // s
    slabs = torch.empty((s * m, n), dtype=torch.float32, device=at.device)
    ops.gemm(at, bt, out=slabs, batch=s, sx=chunk, sw=chunk, sy=m * n, M=m, N=n, K=chunk)
    return K.group_colsum(slabs.reshape(s, m * n)).reshape(m, n)


def _width(m, n, p):
    """Columns of the transposed operands of an [m][n] weight gradient contracting over p rows: p rounded to 8, or a whole number
    of 64-wide K tiles per slice when the contraction is cut (see _splits)."""
    s = _splits(m, n, p)
    return _pad8(p) if s == 1 else s * ((p + s * 64 - 1) // (s * 64)) * 64


def _zeros_operand(rows, width, device, written=0):
    """Operand matrix whose columns [written, width) are zero (transpose_gather itself fills — and zero-pads to a multiple of 8 —
    the first `written` columns of every piece)."""
    out = ops.empty_rows(rows, width, ops.H16(), device)
    if written < width:
        base = out if out._base is None else out._base
        planes = base.shape[1] // width
        for pl in range(planes):
            base[:, pl * width + written:(pl + 1) * width].zero_()
    return out


_recent_transposes = collections.deque(maxlen=2)


def transposed(src, m, n, **kw):
    """transpose_gather into a zero-initialised operand matrix wide enough for the K-slices of the [m][n] weight gradient that
    contracts over it.  A plain transpose is remembered on `src` (weakly, the last two stay alive): the q / k / v projections of an
    attention layer read the same input and run their backward passes one after the other."""
    p = kw.pop("P", None) or src.shape[0]
    width = _width(m, n, p)
    plain = not kw and p == src.shape[0]
    if plain:
        tag = getattr(src, "_mudg_transposed", None)
        if tag is not None and tag[0] == (src._version, width):
            cached = tag[1]()
            if cached is not None:
                return cached
    out = _zeros_operand(src.shape[1], width, src.device, _pad8(p))
    if plain and src.shape[1] % 4 == 0 and src.stride(0) % 4 == 0 and src.data_ptr() % 16 == 0:
        K.transpose_cast_sum(src, out)
    else:
        K.transpose_gather(src, P=p, out=out, **kw)
    if plain:
        _recent_transposes.append(out)
        src._mudg_transposed = ((src._version, width), weakref.ref(out))
    return out


def rows_wgrad_ok(m, c):
    """The row-contracting weight-gradient kernel (mudg_wgrad: no transposed copies) serves the 16-bit builds when the input
    channels come in 64-wide blocks; everything else (the 12-channel input conv, the split-operand builds) takes the transposes."""
    return hip.planes() == 1 and c % 64 == 0 and m % 8 == 0


def grad_rows(dy, rows, sums):
    """(operand rows or None, column sums or None) of an output gradient in one pass (no transposed copy is needed when the weight
    gradient comes from mudg_wgrad)."""
    if sums and dy.shape[1] % 8 == 0 and dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0:
        return K.transpose_cast_sum(dy, None, rows, True)
    return (op(dy) if rows else None), (K.group_colsum(dy)[0] if sums else None)


def grad_forms(dy, m, n, rows, sums):
    """The forms a layer's output gradient dy [P][m] is needed in for an [m][n] weight gradient: (dy^T wide enough for the K-slices,
    operand rows or None, column sums or None) — one pass over dy when its width allows 16-byte accesses."""
    p = dy.shape[0]
    if dy.shape[1] % 8 or dy.stride(0) % 4:
        return transposed(dy, m, n), (op(_pad_cols(dy, _pad8(dy.shape[1]))) if rows else None), (K.group_colsum(dy)[0] if sums else None)
    out = _zeros_operand(dy.shape[1], _width(m, n, p), dy.device, _pad8(p))
    r, s = K.transpose_cast_sum(dy, out, rows, sums)
    return out, r, s


def transposed_taps(x, m, ci, p, taps, mode, geo):
    """[len(taps) * ci][width]: for every tap the transposed copy of the input pixels it read — the right-hand operand of ONE
    weight-gradient GEMM over all taps (N = taps * Cin: more tiles, one launch)."""
    out = _zeros_operand(len(taps) * ci, _width(m, len(taps) * ci, p), x.device, _pad8(p))
    for i, tap in enumerate(taps):
        K.transpose_gather(x, P=p, mode=mode, geo=dict(geo, **tap), out=out[i * ci:(i + 1) * ci])
    return out


# ------------------------------------------------------------------------------------------------ linear
# The rows of one frame of the activation a block is working on: a hint for the library's tile choice (MudgGemmDesc.HW in mode 0: whole
# 288-row tiles per frame -> the 288 x 320 tile, csrc/wgemm.hip).  Set by the UNet walk (train/unet.py) around a block, read when a Linear
# runs — also when activation checkpointing replays the block, which sets it again.
_FRAME_ROWS = [0]


class frame_rows:
    def __init__(self, rows):
        self.rows = int(rows)

    def __enter__(self):
        self.saved = _FRAME_ROWS[0]
        _FRAME_ROWS[0] = self.rows

    def __exit__(self, *exc):
        _FRAME_ROWS[0] = self.saved
        return False


def _hint(rows):
    h = _FRAME_ROWS[0]
    return h if h > 0 and rows % h == 0 else 0


class Linear(torch.autograd.Function):
    """y = x W^T (+ b) (+ residual); x [M][K], W [N][K] (nn.Linear / 1x1 conv weight), fp32 in and out."""

    @staticmethod
    def forward(ctx, x, w, b, residual):
        w2 = w.reshape(w.shape[0], -1)
        xo = op(x)
        # what the weight gradient will read: the operand rows themselves (mudg_wgrad) or the fp32 rows (transposed copies)
        ctx.save_for_backward(xo if rows_wgrad_ok(*w2.shape) else x, w2)
        ctx.wshape, ctx.has_b, ctx.has_r = w.shape, b is not None, residual is not None
        ctx.hint = _hint(x.shape[0])
        return ops.gemm(xo, op(w2), bias=None if b is None else b.float().contiguous(), residual=residual, out_fp32=True, frame_rows=ctx.hint)

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        dy = dy.contiguous()
        n, k = w2.shape
        dx = dw = db = None
        want_b = ctx.has_b and _need(ctx, 2)
        direct = _need(ctx, 1) and rows_wgrad_ok(n, k)
        if direct:
            dyo, db = grad_rows(dy, True, want_b)
        elif _need(ctx, 1):
            dyt, dyo, db = grad_forms(dy, n, k, _need(ctx, 0), want_b)
        else:
            dyo = op(_pad_cols(dy, _pad8(n))) if _need(ctx, 0) else None
            db = K.group_colsum(dy)[0] if want_b else None
        if _need(ctx, 0):
            wt = K.transpose_gather(w2)                                  # [K][N padded]
            dx = ops.gemm(dyo, wt, out_fp32=True, frame_rows=ctx.hint)
        if direct:
            dw = K.wgrad(dyo, x, positions=x.shape[0], m=n, c=k).reshape(ctx.wshape)
        elif _need(ctx, 1):
            dw = wgrad_gemm(dyt, transposed(x, n, k), n, k, x.shape[0]).reshape(ctx.wshape)
        return dx, dw, db, (dy if ctx.has_r else None)


# ------------------------------------------------------------------------------------------------ 3x3 conv
def _conv_weight(w, cin_pad):
    """(Cout, Cin, 3, 3) -> [Cout][tap][Cin padded] fp32 (korder 0)."""
    co, ci = w.shape[:2]
    m = torch.zeros((co, 9, cin_pad), dtype=torch.float32, device=w.device)
    m[:, :, :ci].copy_(w.permute(0, 2, 3, 1).reshape(co, 9, ci))
    return m.reshape(co, 9 * cin_pad)


def _conv_weight_flipped(w, cout_pad):
    """Filter of the input-gradient conv: [Cin][tap'][Cout padded] with tap' the 180-degree rotation of tap."""
    co, ci = w.shape[:2]
    m = torch.zeros((ci, 9, cout_pad), dtype=torch.float32, device=w.device)
    m[:, :, :co].copy_(w.flip(2, 3).permute(1, 2, 3, 0).reshape(ci, 9, co))
    return m.reshape(ci, 9 * cout_pad)


class Conv3x3(torch.autograd.Function):
    """3x3 / pad 1 conv on rows ((f h w), Cin), stride 1 or 2; optional per-row-group bias (the ResBlock's embedding term) and
    residual.  geo = (frames, h, w, stride)."""

    @staticmethod
    def forward(ctx, x, w, b, gbias, residual, geo, rows_per_group):
        frames, h, wd, stride = geo
        co, ci = w.shape[:2]
        cpad = _pad8(ci)
        xo = op(_pad_cols(x, cpad))
        ctx.save_for_backward(xo if rows_wgrad_ok(co, ci) else x, w)
        ctx.geo, ctx.rpg, ctx.flags = geo, rows_per_group, (b is not None, gbias is not None, residual is not None)
        return ops.conv3x3(xo, op(_conv_weight(w, cpad)), frames=frames, hin=h, win=wd, cin=cpad, stride=stride,
                           bias=None if b is None else b.float().contiguous(), gbias=gbias, rows_per_group=rows_per_group or 0,
                           residual=residual, out_fp32=True)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        frames, h, wd, stride = ctx.geo
        has_b, has_g, has_r = ctx.flags
        co, ci = w.shape[:2]
        ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
        dy = dy.contiguous()
        dx = dw = db = dg = None
        want_b = has_b and _need(ctx, 2)
        dyo = None
        direct = _need(ctx, 1) and rows_wgrad_ok(co, ci)
        if direct:
            dyo, db = grad_rows(dy, True, want_b)
        elif _need(ctx, 1):
            dyt, dyo, db = grad_forms(dy, co, 9 * ci, _need(ctx, 0) and stride == 1, want_b)       # [Cout][P padded], rows, sums
        elif want_b:
            db = K.group_colsum(dy)[0]
        if _need(ctx, 0):
            copad = _pad8(co)
            src = dyo
            if src is None or stride != 1:
                src = op(_pad_cols(dy if stride == 1 else K.dilate2x(dy, frames, ho, wo, h, wd), copad))
            dx = ops.conv3x3(src, op(_conv_weight_flipped(w, copad)), frames=frames, hin=h, win=wd, cin=copad, out_fp32=True)
        if direct:
            dw = K.wgrad(dyo, x, positions=frames * ho * wo, m=co, c=ci, taps=9, mode=1,
                         geo=dict(Hin=h, Win=wd, Hout=ho, Wout=wo, stride=stride, pad=1)).reshape(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
        elif _need(ctx, 1):
            p = frames * ho * wo
            g = dict(Hin=h, Win=wd, Hout=ho, Wout=wo, stride=stride, pad=1)
            taps = [dict(dy=ky, dx=kx) for ky in range(3) for kx in range(3)]
            xt = transposed_taps(x, co, ci, p, taps, 1, g)               # [9 Cin][P padded]: what each tap read
            dw = wgrad_gemm(dyt, xt, co, 9 * ci, p).reshape(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
        if has_g and _need(ctx, 3):
            dg = K.group_colsum(dy, rows_per_group=ctx.rpg)
        return dx, dw, db, dg, (dy if has_r else None), None, None


class TConv3(torch.autograd.Function):
    """(3,1,1) temporal conv, pad (1,0,0), on rows ((b t) hw); w (Cout, Cin, 3, 1, 1).  geo = (clips, t, hw)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, geo):
        clips, t, hw = geo
        co, ci = w.shape[:2]
        xo = op(x)
        ctx.save_for_backward(xo if rows_wgrad_ok(co, ci) else x, w)
        ctx.geo, ctx.flags = geo, (b is not None, residual is not None)
        wm = w[:, :, :, 0, 0].permute(0, 2, 1).reshape(co, 3 * ci).contiguous()
        return ops.tconv3(xo, op(wm), clips=clips, t=t, hw=hw, cin=ci, bias=None if b is None else b.float().contiguous(),
                          residual=residual, out_fp32=True)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        clips, t, hw = ctx.geo
        has_b, has_r = ctx.flags
        co, ci = w.shape[:2]
        dy = dy.contiguous()
        dx = dw = db = None
        want_b = has_b and _need(ctx, 2)
        direct = _need(ctx, 1) and rows_wgrad_ok(co, ci)
        if direct:
            dyo, db = grad_rows(dy, True, want_b)
        elif _need(ctx, 1):
            dyt, dyo, db = grad_forms(dy, co, 3 * ci, _need(ctx, 0), want_b)
        else:
            dyo = op(dy) if _need(ctx, 0) else None
            db = K.group_colsum(dy)[0] if want_b else None
        if _need(ctx, 0):
            wf = w[:, :, :, 0, 0].flip(2).permute(1, 2, 0).reshape(ci, 3 * co).contiguous()        # [Cin][tap'][Cout]
            dx = ops.tconv3(dyo, op(wf), clips=clips, t=t, hw=hw, cin=co, out_fp32=True)
        if direct:
            dw = K.wgrad(dyo, x, positions=x.shape[0], m=co, c=ci, taps=3, mode=2,
                         geo=dict(T=t, HW=hw)).reshape(co, 3, ci).permute(0, 2, 1).reshape(co, ci, 3, 1, 1).contiguous()
        elif _need(ctx, 1):
            p = x.shape[0]
            xt = transposed_taps(x, co, ci, p, [dict(dt=d) for d in range(3)], 2, dict(T=t, HW=hw))
            dw = wgrad_gemm(dyt, xt, co, 3 * ci, p).reshape(co, 3, ci).permute(0, 2, 1).reshape(co, ci, 3, 1, 1).contiguous()
        return dx, dw, db, (dy if has_r else None), None


# ------------------------------------------------------------------------------------------------ normalisations, activations
class GroupNorm(torch.autograd.Function):
    """GroupNorm(32) (+ SiLU) over `rows` rows per sample (a frame, or a clip for the temporal blocks)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, samples, rows, eps, silu, groups):
        g, b = gamma.float().contiguous(), beta.float().contiguous()
        y, stat = ops.groupnorm(x, g, b, samples=samples, rows=rows, eps=eps, silu=silu, groups=groups, return_stats=True)
        ctx.save_for_backward(x, g, b, stat)         # (mean, rstd) per (sample, group) as the forward kernels computed them
        ctx.args = (samples, rows, eps, silu, groups)
        return _with_operand(ops.to_f32(y), y)

    @staticmethod
    def backward(ctx, dy):
        x, g, b, stat = ctx.saved_tensors
        samples, rows, eps, silu, groups = ctx.args
        dx, dgamma, dbeta = K.groupnorm_bwd(x, dy.contiguous(), g, b, stat, samples, rows, groups, silu)
        return dx, dgamma, dbeta, None, None, None, None, None


class GroupNormSkip(torch.autograd.Function):
    """x -> (GroupNorm(x) (+ SiLU), x): as LayerNormSkip, for the residual connections around a ResBlock, a temporal conv block and
    a transformer (openaimodel3d.py:236, 279; attention.py:467)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, samples, rows, eps, silu, groups):
        g, b = gamma.float().contiguous(), beta.float().contiguous()
        y, stat = ops.groupnorm(x, g, b, samples=samples, rows=rows, eps=eps, silu=silu, groups=groups, return_stats=True)
        ctx.save_for_backward(x, g, b, stat)
        ctx.args = (samples, rows, eps, silu, groups)
        return _with_operand(ops.to_f32(y), y), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, g, b, stat = ctx.saved_tensors
        samples, rows, eps, silu, groups = ctx.args
        if dy is None:
            return dskip, None, None, None, None, None, None, None
        dx, dgamma, dbeta = K.groupnorm_bwd(x, dy.contiguous(), g, b, stat, samples, rows, groups, silu,
                                            dres=None if dskip is None else dskip.contiguous())
        return dx, dgamma, dbeta, None, None, None, None, None


class LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        g, b = gamma.float().contiguous(), beta.float().contiguous()
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        y = ops.layernorm(x, g, b, eps=eps)
        return _with_operand(ops.to_f32(y), y)

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dx, dg, db = K.layernorm_bwd(x, dy.contiguous(), g, ctx.eps)
        return dx, dg, db, None


class LayerNormSkip(torch.autograd.Function):
    """x -> (LayerNorm(x), x): the second output is what the residual connection around the normalised branch adds back
    (x + f(LN(x)), attention.py:392-400).  Routing the skip through this node lets the backward kernel add the skip's gradient to
    the norm's input gradient as it writes it, instead of autograd accumulating two full-size tensors afterwards."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        g, b = gamma.float().contiguous(), beta.float().contiguous()
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        y = ops.layernorm(x, g, b, eps=eps)
        return _with_operand(ops.to_f32(y), y), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, g = ctx.saved_tensors
        if dy is None:
            return dskip, None, None, None
        dx, dg, db = K.layernorm_bwd(x, dy.contiguous(), g, ctx.eps, dres=None if dskip is None else dskip.contiguous())
        return dx, dg, db, None


class Geglu(torch.autograd.Function):
    """[value | gate] rows -> value * gelu(gate) (attention.py:579-586)."""

    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return K.geglu(h)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return K.geglu(h, dy.contiguous())


class GegluDropout(torch.autograd.Function):
    """GEGLU and the Dropout that follows it in the feed-forward (attention.py:596-606) as one pass in each direction; the forward
    also writes the operand rows the second projection reads.  p = 0: GEGLU alone."""

    @staticmethod
    def forward(ctx, h, p, seed):
        ctx.save_for_backward(h)
        ctx.p, ctx.seed = p, seed
        y, y16 = K.geglu_dropout(h, p, seed, operand=True)
        return _with_operand(y, y16)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return K.geglu_dropout(h, ctx.p, ctx.seed, dy=dy.contiguous()), None, None


def geglu_dropout(mod, h):
    """GEGLU, then the nn.Dropout module `mod` the way it would act now (see dropout())."""
    active = mod is not None and mod.training and mod.p > 0.0
    if h.shape[1] % 8 or h.stride(0) % 4:
        return dropout(mod, Geglu.apply(h))
    return GegluDropout.apply(h, float(mod.p) if active else 0.0, int(torch.randint(0, 2 ** 62, (1,)).item()) if active else 0)


class Silu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return K.silu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.silu(x, dy)


class Dropout(torch.autograd.Function):
    """nn.Dropout(p) in training mode; the mask is a function of (seed, element index) and is regenerated in backward."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        if x.dim() == 2 and x.shape[1] % 8 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
            y, y16 = K.dropout_rows(x, p, seed, operand=True)             # the layer after a Dropout is a conv / projection
            return _with_operand(y, y16)
        return K.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return K.dropout(dy, ctx.p, ctx.seed), None, None


def dropout(mod, x):
    """Apply an nn.Dropout module the way it would act now (identity in eval mode or at p = 0); the seed comes from torch's
    CPU generator, so torch.manual_seed makes a training run reproducible."""
    if mod is None or not mod.training or mod.p <= 0.0:
        return x
    return Dropout.apply(x, float(mod.p), int(torch.randint(0, 2 ** 62, (1,)).item()))


class Add(torch.autograd.Function):
    """a + b on fp32 rows (mudg_axpy_f32) for the places where a residual cannot ride in a GEMM epilogue."""

    @staticmethod
    def forward(ctx, a, b):
        return ops.add_(a.contiguous().clone(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class Upsample2x(torch.autograd.Function):
    """Nearest-2x on rows (openaimodel3d.py:98-103); geo = (frames, h, w)."""

    @staticmethod
    def forward(ctx, x, geo):
        ctx.geo = geo
        return K.upsample2x(x, *geo)

    @staticmethod
    def backward(ctx, dy):
        return K.upsample2x(dy.contiguous(), *ctx.geo, adjoint=True), None


# ------------------------------------------------------------------------------------------------ attention
def _vt(v, batches, nk):
    """V^T per key / value batch as the flash kernels read it: [batches * C][nk padded] operand rows."""
    c = v.shape[1]
    ld = _pad8(nk)
    out = ops.empty_rows(batches * c, ld, ops.H16(), v.device)
    K.transpose_gather(v, P=nk, out=out, batch=batches, src_batch_rows=nk, dst_batch_rows=c)
    return out, out.stride(0)


def _attn_backward_fused(q, k, v, do, groups, nqg, nk, heads, scale, frames, o=None, lse=None):
    """The same gradients from mudg_attention_bwd (16-bit operand builds): no score matrix and no transposed copy in memory."""
    return K.attention_bwd(op(q), op(k), op(v), op(do), frames=frames, heads=heads, nq=nqg * groups // frames, nk=nk,
                           kv_div=frames // groups, scale=scale, o=o, lse=lse)


def _attn_backward_set(q, k, v, do, groups, nqg, nk, heads, scale, frames, o=None, lse=None):
    """Gradients of softmax(scale q k^T) v for one key / value set: `groups` key / value batches, each serving nqg query rows.
    o / lse: the forward output and softmax statistics when the forward pass kept them (single set, 16-bit builds)."""
    if hip.planes() == 1:
        return _attn_backward_fused(q, k, v, do, groups, nqg, nk, heads, scale, frames, o, lse)
    c = q.shape[1]
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    lds, ldq = _pad8(nk), _pad8(nqg)
    dev = q.device
    for h in range(heads):
        hs = slice(h * 64, (h + 1) * 64)
        qh, kh, vh, doh = op(q[:, hs]), op(k[:, hs]), op(v[:, hs]), op(do[:, hs])
        fresh = torch.empty if lds == nk else torch.zeros        # only padding columns need the zeros
        s = fresh((groups * nqg, lds), dtype=torch.float32, device=dev)
        bat = dict(batch=groups, sx=nqg * qh.stride(0), sw=nk * kh.stride(0), sy=nqg * lds, M=nqg, N=nk, K=64)
        ops.gemm(qh, kh, out=s, alpha=scale, **bat)
        K.softmax_f32(s, nk)                                             # s now holds P (the padding columns stay 0)
        dp = fresh(s.shape, dtype=torch.float32, device=dev)
        ops.gemm(doh, vh, out=dp, **dict(bat, sx=nqg * doh.stride(0), sw=nk * vh.stride(0)))
        ds = fresh(s.shape, dtype=torch.float32, device=dev)
        K.softmax_bwd(s, dp, ds, nk, scale)
        # transposed copies, one block of rows per group
        kt = ops.empty_rows(groups * 64, lds, ops.H16(), dev)            # K^T   [64][nk]
        qt = ops.empty_rows(groups * 64, ldq, ops.H16(), dev)            # Q^T   [64][nqg]
        dot = ops.empty_rows(groups * 64, ldq, ops.H16(), dev)           # dO^T  [64][nqg]
        dst = ops.empty_rows(groups * nk, ldq, ops.H16(), dev)           # dS^T  [nk][nqg]
        pt = ops.empty_rows(groups * nk, ldq, ops.H16(), dev)            # P^T   [nk][nqg]
        bt = lambda src, p, out, rows_out: K.transpose_gather(src, P=p, out=out, batch=groups, src_batch_rows=p, dst_batch_rows=rows_out)
        bt(k[:, hs], nk, kt, 64)
        bt(q[:, hs], nqg, qt, 64)
        bt(do[:, hs], nqg, dot, 64)
        bt(ds[:, :nk], nqg, dst, nk)
        bt(s[:, :nk], nqg, pt, nk)
        dso = op(ds)
        dqh = ops.gemm(dso, kt, out_fp32=True, batch=groups, sx=nqg * dso.stride(0), sw=64 * kt.stride(0), sy=nqg * 64, M=nqg, N=64, K=lds,
                       out=torch.empty((groups * nqg, 64), dtype=torch.float32, device=dev))
        dkh = ops.gemm(dst, qt, batch=groups, sx=nk * dst.stride(0), sw=64 * qt.stride(0), sy=nk * 64, M=nk, N=64, K=ldq,
                       out=torch.empty((groups * nk, 64), dtype=torch.float32, device=dev))
        dvh = ops.gemm(pt, dot, batch=groups, sx=nk * pt.stride(0), sw=64 * dot.stride(0), sy=nk * 64, M=nk, N=64, K=ldq,
                       out=torch.empty((groups * nk, 64), dtype=torch.float32, device=dev))
        dq[:, hs].copy_(dqh); dk[:, hs].copy_(dkh); dv[:, hs].copy_(dvh)
    return dq, dk, dv


class Attention(torch.autograd.Function):
    """softmax(scale q k^T) v per (frame, head) on the flash kernels (attention.py:81-144), optionally plus a second key /
    value set with its own softmax (the image tokens of the cross-attention).  q [frames * nq][C]; k, v [(frames / kv_div) * nk][C].
    The backward pass recomputes the probabilities (nothing of size nq x nk is kept from the forward)."""

    @staticmethod
    def forward(ctx, q, k, v, k2, v2, geo):
        frames, heads, nq, nk, kv_div, nk2, kv_div2, scale = geo
        c = q.shape[1]
        ctx.geo = geo
        vt, ldv = _vt(v, frames // kv_div, nk)
        out = ops.empty_rows(frames * nq, c, ops.H16(), q.device)
        kw = {}
        # one key / value set, 16-bit build: the forward kernel leaves its softmax statistics, and the backward pass reads them
        # and the output instead of running a statistics pass over all keys
        ctx.stats = k2 is None and hip.planes() == 1
        if k2 is not None:
            vt2, ldv2 = _vt(v2, frames // kv_div2, nk2)
            kw = dict(k2=op(k2), vt2=vt2, nk2=nk2, ldvt2=ldv2, svt2=c * ldv2, kv_div2=kv_div2)
        elif ctx.stats:
            kw = dict(lse=torch.empty((frames * nq, heads), dtype=torch.float32, device=q.device))
        ops.attention(op(q), op(k), vt, out, frames=frames, heads=heads, nq=nq, nk=nk, ldvt=ldv, svt=c * ldv, kv_div=kv_div, scale=scale, **kw)
        ctx.save_for_backward(q, k, v, *((k2, v2) if k2 is not None else ()), *((out, kw["lse"]) if ctx.stats else ()))
        return _with_operand(ops.to_f32(out), out)

    @staticmethod
    def backward(ctx, do):
        frames, heads, nq, nk, kv_div, nk2, kv_div2, scale = ctx.geo
        saved = ctx.saved_tensors
        q, k, v = saved[:3]
        do = do.contiguous()
        o, lse = (saved[-2], saved[-1]) if ctx.stats else (None, None)
        dq, dk, dv = _attn_backward_set(q, k, v, do, frames // kv_div, kv_div * nq, nk, heads, scale, frames, o, lse)
        dk2 = dv2 = None
        if not ctx.stats and len(saved) == 5:
            dq2, dk2, dv2 = _attn_backward_set(q, saved[3], saved[4], do, frames // kv_div2, kv_div2 * nq, nk2, heads, scale, frames)
            ops.add_(dq, dq2)
        return dq, dk, dv, dk2, dv2, None


class SelfAttention(torch.autograd.Function):
    """Self-attention on the PACKED projection qkv [frames * n][q | k | v] (one GEMM made it, one GEMM takes its gradient back):
    the 16-bit builds' route for attn1 of the spatial blocks.  Keeps the operand rows, the output and the softmax statistics; the
    backward kernels write dq / dk / dv into the column blocks of one packed gradient."""

    @staticmethod
    def forward(ctx, qkv, geo):
        frames, heads, n, scale = geo
        c = heads * 64
        qkv16 = op(qkv)
        q16, k16 = qkv16[:, :c], qkv16[:, c:2 * c]
        vt, ldv = _vt(qkv[:, 2 * c:], frames, n)
        out = ops.empty_rows(frames * n, c, ops.H16(), qkv.device)
        lse = torch.empty((frames * n, heads), dtype=torch.float32, device=qkv.device)
        ops.attention(q16, k16, vt, out, frames=frames, heads=heads, nq=n, nk=n, ldvt=ldv, svt=c * ldv, kv_div=1, scale=scale, lse=lse)
        ctx.save_for_backward(qkv16, out, lse)
        ctx.geo = geo
        return _with_operand(ops.to_f32(out), out)

    @staticmethod
    def backward(ctx, do):
        frames, heads, n, scale = ctx.geo
        qkv16, out, lse = ctx.saved_tensors
        c = heads * 64
        dqkv = torch.empty((frames * n, 3 * c), dtype=torch.float32, device=do.device)
        K.attention_bwd(qkv16[:, :c], qkv16[:, c:2 * c], qkv16[:, 2 * c:], op(do.contiguous()), frames=frames, heads=heads, nq=n, nk=n, kv_div=1,
                        scale=scale, o=out, lse=lse, out=(dqkv[:, :c], dqkv[:, c:2 * c], dqkv[:, 2 * c:]))
        return dqkv, None


class TemporalAttention(torch.autograd.Function):
    """Self-attention over the T frames of every pixel (attention.py:529-576); qkv rows ((b t) hw) x [q | k | v]."""

    @staticmethod
    def forward(ctx, qkv, geo):
        clips, t, hw, heads, scale = geo
        ctx.save_for_backward(qkv)
        ctx.geo = geo
        c = qkv.shape[1] // 3
        out = ops.empty_rows(qkv.shape[0], c, ops.H16(), qkv.device)
        ops.temporal_attention(op(qkv), out, clips=clips, t=t, hw=hw, heads=heads, scale=scale)
        return _with_operand(ops.to_f32(out), out)

    @staticmethod
    def backward(ctx, do):
        (qkv,) = ctx.saved_tensors
        clips, t, hw, heads, scale = ctx.geo
        return K.temporal_attention_bwd(qkv, do.contiguous(), clips, t, hw, heads, scale), None


# ------------------------------------------------------------------------------------------------ layout at the API edge, loss
class ToRows(torch.autograd.Function):
    """(B, C, T, H, W) fp32 -> rows ((b t) h w) x C fp32 (layout only)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        b, c, t, h, w = x.shape
        return x.permute(0, 2, 3, 4, 1).reshape(b * t * h * w, c).contiguous()

    @staticmethod
    def backward(ctx, dy):
        b, c, t, h, w = ctx.shape
        return dy.reshape(b, t, h, w, c).permute(0, 4, 1, 2, 3).contiguous()


class FromRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, shape):
        b, c, t, h, w = shape
        return y.reshape(b, t, h, w, c).permute(0, 4, 1, 2, 3).contiguous()

    @staticmethod
    def backward(ctx, dy):
        b, c, t, h, w = dy.shape
        return dy.permute(0, 2, 3, 4, 1).reshape(b * t * h * w, c).contiguous(), None


class WeightedMSE(torch.autograd.Function):
    """sum_b w[b] * mean((pred_b - target_b)^2): the simple + vlb terms of p_losses (ddpm3d.py:766-787) with their per-sample
    coefficients folded into w.  Returns (loss, per-sample mse)."""

    @staticmethod
    def forward(ctx, pred, target, w):
        w = w.float().contiguous()
        loss_b, grad = K.mse(pred, target, w, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(loss_b)
        total = K.group_colsum(loss_b.reshape(-1, 1), w.reshape(-1, 1))          # sum_b w[b] loss[b], on the device
        return total.reshape(()), loss_b

    @staticmethod
    def backward(ctx, dloss, _):
        (grad,) = ctx.saved_tensors
        b = grad.shape[0]
        scale = dloss.reshape(1).float().expand(b).contiguous()
        return ops.lincomb(grad, grad, scale, torch.zeros(b, dtype=torch.float32, device=grad.device)), None, None
