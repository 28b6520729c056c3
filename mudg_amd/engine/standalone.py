"""Single-block entry points with the reference's tensor layouts, used when a boundary block is called on its own
(the reference's ResBlock / SpatialTransformer / TemporalTransformer are callable modules; tests exercise them one
by one).  Each converts NCHW / NCTHW tensors to channels-last rows, runs the same code the whole-UNet executor runs,
and converts back."""
import torch

from .. import ops
from . import unet as U



def _to_rows(x4):
    """(F, C, H, W) -> rows [F*H*W, C] via the boundary layout kernel (treat F as T of a single clip)."""
    f, c, h, w = x4.shape
    if not x4.is_cuda:
        raise RuntimeError("the MI355X path has no CPU fallback: move inputs to the GPU")
    src = x4 if x4.dtype in (torch.float32, ops.H16()) else x4.float()
    rows = ops.empty_rows(f * h * w, c, ops.H16(), x4.device)
    ops.ncthw_to_rows(src.permute(1, 0, 2, 3).unsqueeze(0), rows, 0)
    return rows


def _from_rows(rows, f, c, h, w, like):
    out = ops.rows_to_ncthw(rows, (1, c, f, h, w), dtype=torch.float32)
    return out[0].permute(1, 0, 2, 3).to(like.dtype)


def _ctx(batch, frames, emb=None):
    ctx = U._Ctx()
    ctx.B, ctx.T, ctx.kv_cache = batch, frames // batch, {}
    ctx.emb = emb
    ctx.text = ctx.img = None
    ctx.n_text = ctx.n_img = ctx.img_div = 0
    return ctx


@torch.no_grad()
def res_block(mod, x, emb, batch_size=None):
    """x ((b t), C, H, W), emb ((b t), E) with identical rows within a clip (as UNetModel.forward builds it)."""
    f, c, h, w = x.shape
    b = batch_size or 1
    if f % b:
        raise ValueError("batch_size must divide the frame count")
    emb_clip = emb.reshape(b, f // b, -1)[:, 0].float().contiguous()
    ctx = _ctx(b, f, emb_clip)
    if not (mod.use_temporal_conv and batch_size):
        saved, mod.use_temporal_conv = mod.use_temporal_conv, False
        try:
            out = U.res_block(mod, _to_rows(x), None, h, w, ctx)
        finally:
            mod.use_temporal_conv = saved
    else:
        out = U.res_block(mod, _to_rows(x), None, h, w, ctx)
    return _from_rows(out, f, mod.out_channels, h, w, x)


@torch.no_grad()
def spatial_transformer(mod, x, context):
    """x (F, C, H, W), context (F, 77 + n_img, D): per-frame text + image tokens as the UNet passes them."""
    f, c, h, w = x.shape
    ctx = _ctx(f, f)                       # every frame its own "clip": per-frame context, T = 1
    U.make_context(mod, ctx, context, 1, x.device)
    out = U.spatial_transformer(mod, _to_rows(x), h, w, ctx)
    return _from_rows(out, f, c, h, w, x)


@torch.no_grad()
def temporal_transformer(mod, x):
    """x (B, C, T, H, W)."""
    b, c, t, h, w = x.shape
    rows = ops.empty_rows(b * t * h * w, c, ops.H16(), x.device)
    ops.ncthw_to_rows(x if x.dtype in (torch.float32, ops.H16()) else x.float(), rows, 0)
    ctx = _ctx(b, b * t)
    out = U.temporal_transformer(mod, rows, h, w, ctx)
    return ops.rows_to_ncthw(out, (b, c, t, h, w), dtype=torch.float32).to(x.dtype)
