"""AutoencoderKL decode on MI355X (reference: lvdm/models/autoencoder.py:104-107, ae_modules.py Decoder.forward
539-578, ResnetBlock 190-210, AttnBlock 52-78, Upsample 123-127; per-frame loop and 1/scale_factor ddpm3d.py:646-667).

Frames are independent, so a whole clip is decoded as one batch of channels-last rows: 3x3 convs are implicit
GEMMs (nearest-2x upsampling fused into the loader), GroupNorm(eps 1e-6)+swish is the fused norm kernel, and the
single-head d=C attention of the mid block is two batched MFMA GEMMs around a row softmax (scores in fp32).
The 1/scale_factor latent scaling is folded into the 1x1 post_quant_conv GEMM's alpha.
"""
import torch
import torch.nn as nn

from .. import ops
from . import packing as pk
from . import unet as U

MAX_SCORE_BYTES = 8 << 30      # fp32 attention scores held at once (frames are chunked beyond this)


def _gn(mod, x, frames, hw, swish):
    return ops.groupnorm(x, pk.f32(mod, "weight"), pk.f32(mod, "bias"), samples=frames, rows=hw, eps=mod.eps,
                         silu=swish, groups=mod.num_groups)


def _conv3x3(mod, x, frames, h, w, upsample=False, residual=None, stream=False, stride=1, pad=1, fp32=False):
    """stream=True: the output is a residual-stream tensor, kept in ops.STREAM() (same policy as the UNet executor).  Every conv
    here except the nearest-2x ones (16-wave kernel) also writes the partial sums of the GroupNorm that follows it."""
    wmat, cpad, korder = pk.conv3x3(mod)
    return ops.conv3x3(x, wmat, frames=frames, hin=h, win=w, cin=cpad, upsample=upsample, bias=pk.f32(mod, "bias"),
                       residual=residual, out_stream=stream, out_fp32=fp32, korder=korder, stride=stride, pad=pad,
                       stats=not (upsample or fp32))


def resnet_block(mod, x, frames, h, w):
    a = _conv3x3(mod.conv1, _gn(mod.norm1, x, frames, h * w, True), frames, h, w)
    a = _gn(mod.norm2, a, frames, h * w, True)
    skip = x
    if mod.in_channels != mod.out_channels:
        skip = ops.gemm(ops.cast_bf16(x), pk.linear(mod.nin_shortcut), bias=pk.f32(mod.nin_shortcut, "bias"),
                        out_stream=True)
    return _conv3x3(mod.conv2, a, frames, h, w, residual=skip, stream=True)


def attn_block(mod, x, frames, hw):
    """x + proj_out(softmax(q k^T / sqrt(C)) v), one head of width C over the hw positions of each frame."""
    c = mod.in_channels
    if hw % 8:
        raise NotImplementedError(f"VAE attention needs h*w divisible by 8 (got {hw})")
    hn = _gn(mod.norm, x, frames, hw, False)
    wqk = pk.cached(mod, "qk", (mod.q.weight, mod.k.weight),
                    lambda: pk.operand(torch.cat([mod.q.weight.detach().reshape(c, c), mod.k.weight.detach().reshape(c, c)], 0)))
    bqk = pk.cached(mod, "bqk", (mod.q.bias, mod.k.bias),
                    lambda: torch.cat([mod.q.bias.detach(), mod.k.bias.detach()]).float().contiguous())
    qk = ops.gemm(hn, wqk, bias=bqk)                                   # [frames*hw, 2C]
    ldv = (hw + 7) // 8 * 8
    vt = ops.empty_rows(frames * c, ldv, ops.H16(), x.device)               # V^T per frame (bias added after P @ V)
    ops.gemm(pk.linear(mod.v), hn, out=vt, batch=frames, sx=0, sw=hw * hn.stride(0), sy=c * vt.stride(0), M=c, N=hw, K=c)
    att = ops.empty_rows(frames * hw, c, ops.H16(), x.device)
    per = max(1, min(frames, MAX_SCORE_BYTES // (hw * hw * 4)))
    scores = torch.empty((per * hw, hw), dtype=torch.float32, device=x.device)
    probs = ops.empty_rows(per * hw, hw, ops.H16(), x.device)
    q, k = qk[:, :c], qk[:, c:]
    for f0 in range(0, frames, per):
        n = min(per, frames - f0)
        rows = slice(f0 * hw, (f0 + n) * hw)
        ops.gemm(q[rows], k[rows], out=scores, out_fp32=True, alpha=float(int(c) ** (-0.5)), batch=n,
                 sx=hw * qk.stride(0), sw=hw * qk.stride(0), sy=hw * hw, M=hw, N=hw, K=c, ldy=hw)
        ops.softmax_rows(scores[:n * hw], probs[:n * hw])
        ops.gemm(probs, vt[f0 * c:(f0 + n) * c], out=att[rows], bias=pk.f32(mod.v, "bias"), batch=n, sx=hw * probs.stride(0),
                 sw=c * vt.stride(0), sy=hw * att.stride(0), M=hw, N=c, K=hw)
    return ops.gemm(att, pk.linear(mod.proj_out), bias=pk.f32(mod.proj_out, "bias"), residual=x, out_stream=True)


def decoder_rows(dec, x, frames, h, w):
    """Decoder.forward on rows [frames*h*w, z-padded]; returns (rows [frames*H*W, out_ch], H, W)."""
    x = _conv3x3(dec.conv_in, x, frames, h, w, stream=True)
    x = resnet_block(dec.mid.block_1, x, frames, h, w)
    if not isinstance(dec.mid.attn_1, nn.Identity):
        x = attn_block(dec.mid.attn_1, x, frames, h * w)
    x = resnet_block(dec.mid.block_2, x, frames, h, w)
    for lvl in reversed(range(dec.num_resolutions)):
        level = dec.up[lvl]
        for i in range(dec.num_res_blocks + 1):
            x = resnet_block(level.block[i], x, frames, h, w)
            if len(level.attn) > 0:
                x = attn_block(level.attn[i], x, frames, h * w)
        if lvl != 0:
            x = U.upsample_conv(level.upsample.conv, ops.cast_bf16(x), frames, h, w)
            h, w = 2 * h, 2 * w
    x = _gn(dec.norm_out, x, frames, h * w, True)
    return _conv3x3(dec.conv_out, x, frames, h, w, fp32=True), h, w          # the decoded pixels leave in fp32


def encoder_rows(enc, x, frames, h, w):
    """Encoder.forward (ae_modules.py:433-463) on rows [frames*h*w, in-padded]; returns (rows, h/8.., w/8..)."""
    x = _conv3x3(enc.conv_in, x, frames, h, w, stream=True)
    for lvl in range(enc.num_resolutions):
        level = enc.down[lvl]
        for i in range(enc.num_res_blocks):
            x = resnet_block(level.block[i], x, frames, h, w)
            if len(level.attn) > 0:
                x = attn_block(level.attn[i], x, frames, h * w)
        if lvl != enc.num_resolutions - 1:
            # Downsample: zero-pad bottom/right by one, 3x3 stride 2, no other padding (ae_modules.py:102-107)
            x = _conv3x3(level.downsample.conv, ops.cast_bf16(x), frames, h, w, stream=True, stride=2, pad=0)
            h, w = (h + 1 - 3) // 2 + 1, (w + 1 - 3) // 2 + 1
    x = resnet_block(enc.mid.block_1, x, frames, h, w)
    if not isinstance(enc.mid.attn_1, nn.Identity):
        x = attn_block(enc.mid.attn_1, x, frames, h * w)
    x = resnet_block(enc.mid.block_2, x, frames, h, w)
    x = _gn(enc.norm_out, x, frames, h * w, True)
    return _conv3x3(enc.conv_out, x, frames, h, w), h, w


@torch.no_grad()
def encode_moments(ae, x, max_frames=8):
    """AutoencoderKL.encode up to the posterior parameters: x (N, 3, H, W) -> moments (N, 2*embed, H/8, W/8) fp32
    (encoder + 1x1 quant_conv).  Frames are independent and processed in batches."""
    x = _check(x).contiguous()
    n, c, h, w = x.shape
    cin = (c + 7) // 8 * 8
    qc = ae.quant_conv
    out = None
    for n0 in range(0, n, max_frames):
        nb = min(max_frames, n - n0)
        rows = ops.empty_rows(nb * h * w, cin, ops.H16(), x.device)
        ops.ncthw_to_rows(x[n0:n0 + nb].unsqueeze(2), rows, 0)
        if cin > c:
            ops.zero_channels(rows, c, cin)
        y, hh, ww = encoder_rows(ae.encoder, rows, nb, h, w)
        mom = ops.gemm(y, pk.linear(qc), bias=pk.f32(qc, "bias"), out_fp32=True)
        if out is None:
            out = torch.empty((n, mom.shape[1], 1, hh, ww), dtype=torch.float32, device=x.device)
        ops.rows_to_ncthw(mom, (nb, mom.shape[1], 1, hh, ww), out=out[n0:n0 + nb])
    return out[:, :, 0]


def _check(z):
    if not z.is_cuda:
        raise RuntimeError("AutoencoderKL.decode: latents must be on the GPU; the MI355X path has no CPU fallback")
    return z if z.dtype in (torch.float32, ops.H16()) else z.float()


def _decode_rows(ae, rows_z, frames, h, w, inv_scale):
    """rows_z [frames*h*w, zc] bf16 latents -> decoded rows; post_quant_conv with the latent scale as GEMM alpha."""
    pq = ae.post_quant_conv
    zc = pq.weight.shape[0]
    cpad = (zc + 7) // 8 * 8
    lat = ops.empty_rows(rows_z.shape[0], cpad, ops.H16(), rows_z.device)
    if cpad > zc:
        ops.zero_channels(lat, zc, cpad)
    wpq = pk.cached(pq, "wpad", (pq.weight,), lambda: pk.operand(torch.nn.functional.pad(
        pq.weight.detach().reshape(zc, -1), (0, rows_z.shape[1] - pq.weight.shape[1]))))
    ops.gemm(rows_z, wpq, out=lat, bias=pk.f32(pq, "bias"), alpha=inv_scale, N=zc)
    return decoder_rows(ae.decoder, lat, frames, h, w)


@torch.no_grad()
def decode_latents(ae, z, inv_scale=1.0, perframe=True, max_frames=16):
    """(B, C, T, h, w) -> (B, 3, T, 8h, 8w) or (N, C, h, w) -> (N, 3, 8h, 8w); fp32 output like the reference.
    Frames are decoded in batches of up to `max_frames` (they are independent: perframe_ae only trades launches for
    memory in the reference); the layout kernels read / write frame windows of the NCTHW tensors in place."""
    z = _check(z).contiguous()
    five = z.dim() == 5
    if not five:
        z = z.unsqueeze(2)                           # (N, C, 1, h, w): each image a one-frame clip
    b, c, t, h, w = z.shape
    cin = (c + 7) // 8 * 8
    out = None
    # work list of (first clip, clips, first frame, frames) with clips * frames <= max_frames
    if t >= max_frames:
        jobs = [(b0, 1, t0, min(max_frames, t - t0)) for b0 in range(b) for t0 in range(0, t, max_frames)]
    else:
        per = max(1, max_frames // t)
        jobs = [(b0, min(per, b - b0), 0, t) for b0 in range(0, b, per)]
    for b0, nb, t0, nt in jobs:
        rows = ops.empty_rows(nb * nt * h * w, cin, ops.H16(), z.device)
        ops.ncthw_to_rows(z[b0:b0 + nb], rows, 0, t0=t0, frames=nt)
        if cin > c:
            ops.zero_channels(rows, c, cin)
        y, H, W = _decode_rows(ae, rows, nb * nt, h, w, inv_scale)
        if out is None:
            out = torch.empty((b, y.shape[1], t, H, W), dtype=torch.float32, device=z.device)
        ops.rows_to_ncthw(y, (nb, y.shape[1], t, H, W), out=out[b0:b0 + nb], t0=t0, frames=nt)
    return out if five else out[:, :, 0]


def decode(ae, z):
    """AutoencoderKL.decode: z (N, zc, h, w), no latent scaling."""
    return decode_latents(ae, z, 1.0)


def decoder_forward(dec, z):
    """Decoder.forward on an already post-quant-convolved z (N, zc, h, w)."""
    z = _check(z).contiguous()
    n, c, h, w = z.shape
    cin = (c + 7) // 8 * 8
    rows = ops.empty_rows(n * h * w, cin, ops.H16(), z.device)
    ops.ncthw_to_rows(z.unsqueeze(2), rows, 0)
    if cin > c:
        ops.zero_channels(rows, c, cin)
    y, H, W = decoder_rows(dec, rows, n, h, w)
    return ops.rows_to_ncthw(y, (n, y.shape[1], 1, H, W), dtype=torch.float32)[:, :, 0]
