"""The training step on the GPU: every autograd Function of mudg_amd/train/functions.py — forward AND backward are
HIP kernels — against torch autograd of the same op on the CPU in fp64, then the whole UNet: loss and the gradient of EVERY
parameter of p_losses (v-prediction MSE, ddpm3d.py:741-802) against autograd of the CPU oracle (oracle/unet.py restates the
reference's forward; its functions are plain differentiable torch), and one AdamW step against torch.optim.AdamW.

Tolerances follow the operand mode: the forward and both backward contractions round their operands (2^-9 bf16 / 2^-12 fp16;
16 / 24 bits in the precision modes), so a gradient carries a few operand roundings; bounds are printed with the measurement."""
import copy

import pytest
import torch
import torch.nn.functional as F

from helpers import golden, rel_l2, seeded_sd, unet_inputs

from mudg_amd import hip as _hip

pytestmark = pytest.mark.gpu
MODE = _hip.operand_name()
TOL = {"bf16": 1.5e-2, "fp16": 2e-3, "bf16x3": 2e-4, "bf16x6": 2e-5}[MODE]          # one op, forward value or gradient
TOL_NET = {"bf16": 8e-2, "fp16": 1e-2, "bf16x3": 1e-3, "bf16x6": 1e-4}[MODE]          # gradients through the whole UNet


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def check(name, got, want, tol=TOL):
    err = rel_l2(got, want)
    print(f"[{MODE}] {name}: rel-L2 {err:.3e} (bound {tol:g})")
    assert err < tol, (name, err)


def run_both(fn_hip, fn_ref, tensors, cuda, grad_seed=99):
    """Forward + backward of `fn_hip` on GPU copies and of `fn_ref` on fp64 CPU copies of `tensors` (dict name -> tensor);
    returns ((y, grads), (y_ref, grads_ref))."""
    g = {k: v.to(cuda).requires_grad_(True) for k, v in tensors.items()}
    r = {k: v.double().requires_grad_(True) for k, v in tensors.items()}
    y, yr = fn_hip(**g), fn_ref(**r)
    dy = rnd(*yr.shape, seed=grad_seed)
    y.backward(dy.to(cuda))
    yr.backward(dy.double())
    return (y, {k: v.grad for k, v in g.items()}), (yr, {k: v.grad for k, v in r.items()})


def compare(name, hip_out, ref_out, tol=TOL):
    (y, gh), (yr, gr) = hip_out, ref_out
    check(f"{name} forward", y, yr, tol)
    for k in gr:
        check(f"{name} d{k}", gh[k], gr[k], tol)


def test_linear_conv_and_temporal_conv_gradients(cuda):
    from mudg_amd.train import functions as Fn
    M, Kd, N = 200, 96, 72
    t = dict(x=rnd(M, Kd, seed=1), w=rnd(N, Kd, seed=2, scale=0.1), b=rnd(N, seed=3), r=rnd(M, N, seed=4))
    compare("linear", *run_both(lambda x, w, b, r: Fn.Linear.apply(x, w, b, r), lambda x, w, b, r: F.linear(x, w, b) + r, t, cuda))
    # many rows: the weight-gradient contraction is cut into K-slices (functions._splits) summed in a fixed order
    M = 4100
    t = dict(x=rnd(M, Kd, seed=1), w=rnd(N, Kd, seed=2, scale=0.1), b=rnd(N, seed=3))
    assert Fn._splits(N, Kd, M) > 1
    compare("linear (K-sliced dW)", *run_both(lambda x, w, b: Fn.Linear.apply(x, w, b, None), lambda x, w, b: F.linear(x, w, b), t, cuda))
    # input widths in 64-channel blocks take the row-contracting weight-gradient kernel (mudg_wgrad) in the 16-bit builds: one
    # slice, many slices with a ragged last K-step, an output narrower / wider than a tile
    for M, Kd, N in ((200, 128, 72), (4100, 128, 72), (1000, 320, 328)):
        t = dict(x=rnd(M, Kd, seed=1), w=rnd(N, Kd, seed=2, scale=0.1), b=rnd(N, seed=3))
        compare(f"linear {M} x {Kd} -> {N}", *run_both(lambda x, w, b: Fn.Linear.apply(x, w, b, None), lambda x, w, b: F.linear(x, w, b), t, cuda))
    # 3x3 conv: stride 1 with the embedding (row-group) bias and a residual, then stride 2; Cin = 12 exercises the channel padding;
    # Cin = 192: an output tile of the weight gradient straddles two taps; odd sizes under stride 2
    # (the last three: image rows of 64 and 128 pixels and a 16-pixel one — the affine tap paths of mudg_wgrad, where a 64-position
    # K-step is part of one image row or spans whole rows)
    for stride, ci, co, h, wd in ((1, 64, 40, 6, 8), (2, 12, 24, 6, 8), (1, 64, 64, 24, 32), (1, 192, 136, 10, 12), (2, 64, 72, 9, 11),
                                 (1, 64, 72, 5, 64), (1, 64, 40, 3, 128), (1, 128, 64, 8, 16), (1, 64, 72, 16, 64)):     # the last: several slices
        frames = 4
        ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
        t = dict(x=rnd(frames * h * wd, ci, seed=1), w=rnd(co, ci, 3, 3, seed=2, scale=0.1), b=rnd(co, seed=3),
                 e=rnd(2, co, seed=4), r=rnd(frames * ho * wo, co, seed=5))

        def ref(x, w, b, e, r):
            y = F.conv2d(x.reshape(frames, h, wd, ci).permute(0, 3, 1, 2), w, b, stride=stride, padding=1)
            y = y + e.repeat_interleave(frames // 2, 0)[:, :, None, None]
            return y.permute(0, 2, 3, 1).reshape(-1, co) + r
        compare(f"conv3x3 stride {stride}", *run_both(
            lambda x, w, b, e, r: Fn.Conv3x3.apply(x, w, b, e, r, (frames, h, wd, stride), (frames // 2) * ho * wo), ref, t, cuda))
    for clips, tt, hw, c in ((2, 5, 12, 64), (2, 7, 50, 128), (1, 4, 9, 24), (2, 5, 64, 64), (1, 3, 192, 128), (2, 8, 256, 64)):      # the last three: frames of whole K-steps, the last in several slices
        t = dict(x=rnd(clips * tt * hw, c, seed=1), w=rnd(c, c, 3, 1, 1, seed=2, scale=0.1), b=rnd(c, seed=3))

        def tref(x, w, b):
            y = F.conv3d(x.reshape(clips, tt, hw, 1, c).permute(0, 4, 1, 2, 3), w, b, padding=(1, 0, 0))
            return y.permute(0, 2, 3, 4, 1).reshape(-1, c) + x
        compare(f"tconv3 {c} channels", *run_both(lambda x, w, b: Fn.TConv3.apply(x, w, b, x, (clips, tt, hw)), tref, t, cuda))


def test_norm_geglu_resampling_and_loss_gradients(cuda):
    from mudg_amd.train import functions as Fn
    samples, rows, c = 3, 50, 64
    t = dict(x=rnd(samples * rows, c, seed=1) * 2 + 0.5, g=1 + 0.2 * rnd(c, seed=2), b=0.2 * rnd(c, seed=3))
    for silu in (True, False):
        def ref(x, g, b):
            y = F.group_norm(x.reshape(samples, rows, c).transpose(1, 2), 32, g, b, 1e-5)
            return (F.silu(y) if silu else y).transpose(1, 2).reshape(-1, c)
        compare(f"groupnorm silu={silu}", *run_both(lambda x, g, b: Fn.GroupNorm.apply(x, g, b, samples, rows, 1e-5, silu, 32), ref, t, cuda))
    t = dict(x=rnd(77, 320, seed=1) * 3, g=1 + 0.2 * rnd(320, seed=2), b=0.2 * rnd(320, seed=3))
    compare("layernorm", *run_both(lambda x, g, b: Fn.LayerNorm.apply(x, g, b, 1e-5), lambda x, g, b: F.layer_norm(x, (320,), g, b, 1e-5), t, cuda))
    t = dict(h=rnd(60, 256, seed=1))
    compare("geglu", *run_both(lambda h: Fn.Geglu.apply(h), lambda h: h[:, :128] * F.gelu(h[:, 128:]), t, cuda), tol=1e-5)
    # GEGLU + the feed-forward's Dropout as one pass each way: p = 0 is GEGLU; p > 0 keeps ~(1 - p) of the outputs, scaled, and the
    # backward pass regenerates exactly the forward's mask
    t = dict(h=rnd(60, 256, seed=1))
    compare("geglu + dropout(0)", *run_both(lambda h: Fn.GegluDropout.apply(h, 0.0, 0), lambda h: h[:, :128] * F.gelu(h[:, 128:]), t, cuda), tol=1e-5)
    hh = rnd(400, 512, seed=2).to(cuda).requires_grad_()
    y = Fn.GegluDropout.apply(hh, 0.25, 1234)
    keep = (y != 0).double().cpu()
    assert abs(float(keep.mean()) - 0.75) < 0.02 and torch.equal(y, Fn.GegluDropout.apply(hh, 0.25, 1234)) and not torch.equal(y, Fn.GegluDropout.apply(hh, 0.25, 99))
    dyy = rnd(400, 256, seed=3)
    y.backward(dyy.to(cuda))
    h64 = hh.detach().cpu().double().requires_grad_()
    ref = h64[:, :256] * F.gelu(h64[:, 256:]) * keep / 0.75
    ref.backward(dyy.double())
    check("geglu + dropout(0.25) forward", y, ref, 1e-5)
    check("geglu + dropout(0.25) dh", hh.grad, h64.grad, 1e-5)
    if getattr(y, "_mudg_operand", None) is not None:                   # the operand rows handed to the second projection
        from mudg_amd import ops as _ops
        check("geglu + dropout operand rows", _ops.to_f32(Fn.op(y)), y, 5e-3 if MODE in ("bf16", "fp16") else 1e-5)
    t = dict(x=rnd(2 * 3 * 5, 16, seed=1))
    compare("upsample2x", *run_both(lambda x: Fn.Upsample2x.apply(x, (2, 3, 5)),
                                    lambda x: F.interpolate(x.reshape(2, 3, 5, 16).permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
                                    .permute(0, 2, 3, 1).reshape(-1, 16), t, cuda), tol=1e-6)
    t = dict(x=rnd(40, 8, seed=1))
    compare("silu", *run_both(lambda x: Fn.Silu.apply(x), lambda x: F.silu(x), t, cuda), tol=1e-5)
    # weighted MSE: loss = sum_b w_b mean_b((p - t)^2)
    p, tg, w = rnd(3, 4, 2, 5, 5, seed=1), rnd(3, 4, 2, 5, 5, seed=2), torch.tensor([0.5, 0.2, 0.3])
    ph = p.to(cuda).requires_grad_(True)
    loss, per = Fn.WeightedMSE.apply(ph, tg.to(cuda), w.to(cuda))
    (loss * 2.0).backward()
    pr = p.double().requires_grad_(True)
    lr = (w.double() * ((pr - tg.double()) ** 2).mean(dim=(1, 2, 3, 4))).sum()
    (lr * 2.0).backward()
    check("mse loss", loss, lr, 1e-6); check("mse per sample", per, ((p - tg) ** 2).mean(dim=(1, 2, 3, 4)), 1e-6); check("mse grad", ph.grad, pr.grad, 1e-6)
    # dropout: inverted scaling, the same mask forward and backward, and a sensible keep rate
    x = torch.ones(4096, 64, device=cuda, requires_grad=True)
    y = Fn.Dropout.apply(x, 0.25, 1234)
    y.backward(torch.ones_like(y))
    keep = (y != 0).float().mean().item()
    assert torch.equal(y.detach(), x.grad) and abs(keep - 0.75) < 0.01 and torch.allclose(y[y != 0], torch.tensor(1 / 0.75, device=cuda))


def _attn_ref(q, k, v, frames, heads, nq, nk, kv_div, scale):
    c = q.shape[1]
    qh = q.reshape(frames, nq, heads, 64).transpose(1, 2)
    kh = k.reshape(frames // kv_div, nk, heads, 64).transpose(1, 2).repeat_interleave(kv_div, 0)
    vh = v.reshape(frames // kv_div, nk, heads, 64).transpose(1, 2).repeat_interleave(kv_div, 0)
    return (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).transpose(1, 2).reshape(frames * nq, c)


def test_attention_gradients_self_cross_two_sets_and_temporal(cuda):
    from mudg_amd.train import functions as Fn
    frames, heads, n = 4, 2, 40
    c = heads * 64
    t = dict(q=rnd(frames * n, c, seed=1), k=rnd(frames * n, c, seed=2), v=rnd(frames * n, c, seed=3))
    compare("self-attention", *run_both(lambda q, k, v: Fn.Attention.apply(q, k, v, None, None, (frames, heads, n, n, 1, 0, 1, 0.125)),
                                        lambda q, k, v: _attn_ref(q, k, v, frames, heads, n, n, 1, 0.125), t, cuda))
    # several query and key tiles with ragged tails on both sides (the fused backward walks 64-row tiles under 128-row workgroups)
    m = 333
    t = dict(q=rnd(2 * m, c, seed=6), k=rnd(2 * m, c, seed=7), v=rnd(2 * m, c, seed=8))
    compare("self-attention, 333 tokens", *run_both(lambda q, k, v: Fn.Attention.apply(q, k, v, None, None, (2, heads, m, m, 1, 0, 1, 0.125)),
                                                    lambda q, k, v: _attn_ref(q, k, v, 2, heads, m, m, 1, 0.125), t, cuda))
    # the long-sequence forward kernels (64 queries per wave: LDS-DMA staged when the keys come in whole 64-tiles, register-staged
    # otherwise) hand their softmax statistics to the backward pass like the short one does
    for m, mk in ((576, 576), (520, 300)):
        t = dict(q=rnd(m, c, seed=12), k=rnd(mk, c, seed=13), v=rnd(mk, c, seed=14))
        compare(f"attention {m} queries x {mk} keys", *run_both(lambda q, k, v: Fn.Attention.apply(q, k, v, None, None, (1, heads, m, mk, 1, 0, 1, 0.125)),
                                                                lambda q, k, v: _attn_ref(q, k, v, 1, heads, m, mk, 1, 0.125), t, cuda))
    t = dict(q=rnd(4 * 200, c, seed=9), k=rnd(2 * 150, c, seed=10), v=rnd(2 * 150, c, seed=11))
    compare("two frames per key batch, 200 queries, 150 keys", *run_both(
        lambda q, k, v: Fn.Attention.apply(q, k, v, None, None, (4, heads, 200, 150, 2, 0, 1, 0.125)),
        lambda q, k, v: _attn_ref(q, k, v, 4, heads, 200, 150, 2, 0.125), t, cuda))
    if MODE in ("bf16", "fp16"):      # the packed route of the 16-bit builds: q | k | v in one matrix, one packed gradient back
        for fr, m in ((2, 40), (1, 576)):
            packed = dict(qkv=rnd(fr * m, 3 * c, seed=15))
            compare(f"packed self-attention {m} tokens", *run_both(
                lambda qkv: Fn.SelfAttention.apply(qkv, (fr, heads, m, 0.125)),
                lambda qkv: _attn_ref(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], fr, heads, m, m, 1, 0.125), packed, cuda))
    # no atomics, fixed summation order: the same inputs give the same bits
    q1, k1, v1 = (x.to(cuda).requires_grad_() for x in (t["q"], t["k"], t["v"]))
    grads = []
    for _ in range(2):
        out = Fn.Attention.apply(q1, k1, v1, None, None, (4, heads, 200, 150, 2, 0, 1, 0.125))
        grads.append(torch.autograd.grad(out.square().sum(), (q1, k1, v1)))
    assert all(torch.equal(a, b) for a, b in zip(*grads)), "attention backward is not bit-reproducible"
    T = 2      # text keys shared by the T frames of a clip (77 tokens, not a multiple of 8) + per-frame image keys, summed
    t = dict(q=rnd(frames * n, c, seed=1), k=rnd(frames // T * 77, c, seed=2), v=rnd(frames // T * 77, c, seed=3),
             k2=rnd(frames * 16, c, seed=4), v2=rnd(frames * 16, c, seed=5))
    compare("text + image cross-attention", *run_both(
        lambda q, k, v, k2, v2: Fn.Attention.apply(q, k, v, k2, v2, (frames, heads, n, 77, T, 16, 1, 0.125)),
        lambda q, k, v, k2, v2: _attn_ref(q, k, v, frames, heads, n, 77, T, 0.125) + _attn_ref(q, k2, v2, frames, heads, n, 16, 1, 0.125), t, cuda))
    clips, tt, hw = 2, 6, 10
    t = dict(qkv=rnd(clips * tt * hw, 3 * c, seed=1))

    def tref(qkv):
        x = qkv.reshape(clips, tt, hw, 3, heads, 64).permute(3, 0, 2, 4, 1, 5)            # (3, b, s, h, t, d)
        o = torch.softmax(x[0] @ x[1].transpose(-1, -2) * 0.125, -1) @ x[2]               # (b, s, h, t, d)
        return o.permute(0, 3, 1, 2, 4).reshape(clips * tt * hw, c)
    compare("temporal attention", *run_both(lambda qkv: Fn.TemporalAttention.apply(qkv, (clips, tt, hw, heads, 0.125)), tref, t, cuda))


def _tiny_model(cuda):
    from helpers import cfgs
    from lvdm.models.ddpm3d import LatentVisualDiffusion
    g = golden("unet_b.pt")
    ident = {"target": "torch.nn.Identity"}
    model = LatentVisualDiffusion(
        img_cond_stage_config=ident, image_proj_stage_config=ident, cond_stage_config=ident,
        first_stage_config={"target": "torch.nn.Identity"},
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": g["cfg"]}, **cfgs.DIFFUSION)
    sd = seeded_sd(g["param_shapes"], g["seed"], g["checksum"])
    model.model.diffusion_model.load_state_dict(sd, strict=True)
    model = model.to(cuda).train()
    for m in model.modules():                 # parity needs the same function on both sides: no random masks
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model, g, sd


def test_p_losses_loss_and_every_parameter_gradient_match_autograd_of_the_cpu_oracle(cuda):
    from oracle import unet as o_unet
    model, g, sd = _tiny_model(cuda)
    shp = g["shape"]
    x, ctx = unet_inputs(g["cfg"], shp, g["seed"])                        # x: (B, 12, T, H, W) = latents + c_concat
    x_start, concat = x[:, :4].contiguous(), x[:, 4:].contiguous()
    noise = rnd(*x_start.shape, seed=7)
    t = torch.tensor([999, 420, 17])[:shp["B"]]
    label = torch.tensor([0, 500, 1])[:shp["B"], None]
    fs = torch.full((shp["B"],), 10)
    cond = {"c_crossattn": [ctx.to(cuda)], "c_concat": [concat.to(cuda)]}
    loss, info = model.p_losses(x_start.to(cuda), cond, t.to(cuda), noise=noise.to(cuda), class_label=label.to(cuda), fs=fs.to(cuda))
    loss.backward()
    unet = model.model.diffusion_model
    # ---- the same step on the CPU oracle, differentiated by torch.autograd
    ref_sd = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    sac, s1m = model.sqrt_alphas_cumprod.cpu()[t], model.sqrt_one_minus_alphas_cumprod.cpu()[t]
    bc = lambda v: v[:, None, None, None, None]
    x_noisy = bc(sac) * x_start + bc(s1m) * noise
    target = bc(sac) * noise - bc(s1m) * x_start
    forward = o_unet.unet_forward.__wrapped__                               # the restatement without its no_grad wrapper
    pred = forward(ref_sd, g["cfg"], torch.cat([x_noisy, concat], 1), t, label[:, 0], ctx, fs)
    want = ((pred - target) ** 2).mean()
    want.backward()
    check("p_losses loss", loss, want, TOL_NET)
    assert abs(float(info["train/loss_simple"]) - float(want.detach())) < TOL_NET * float(want.detach())
    worst, missing = 0.0, []
    num = den = 0.0
    for name, p in unet.named_parameters():
        gr = ref_sd[name].grad
        if p.grad is None:
            missing.append(name)
            continue
        d = (p.grad.double().cpu() - gr).norm().item()
        num += d * d; den += gr.norm().item() ** 2
        if gr.norm() > 0:
            worst = max(worst, d / gr.norm().item())
    assert not missing, missing[:5]
    total = (num / den) ** 0.5
    print(f"[{MODE}] UNet parameter gradients vs oracle autograd: {len(list(unet.parameters()))} tensors, overall rel-L2 {total:.3e}, "
          f"worst single tensor {worst:.3e} (bound {TOL_NET:g} overall)")
    assert total < TOL_NET and worst < 10 * TOL_NET


def test_activation_checkpointing_replays_blocks_and_gives_the_same_gradients(cuda):
    """use_checkpoint (lvdm/common.py:81-93): block forwards are replayed in backward; with fixed-order kernels the gradients
    are bit-identical to the stored-activation run, dropout included (the mask seed is replayed with the RNG state)."""
    model, g, _ = _tiny_model(cuda)
    unet = model.model.diffusion_model
    for m in unet.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.1
    shp = g["shape"]
    x, ctx = unet_inputs(g["cfg"], shp, g["seed"])
    args = (x[:, :4].contiguous().to(cuda), {"c_crossattn": [ctx.to(cuda)], "c_concat": [x[:, 4:].contiguous().to(cuda)]},
            torch.tensor([700, 420, 100])[:shp["B"]].to(cuda))
    kw = dict(noise=rnd(shp["B"], 4, shp["T"], shp["H"], shp["W"], seed=3).to(cuda), class_label=torch.tensor([0, 500, 1])[:shp["B"], None].to(cuda),
              fs=torch.full((shp["B"],), 10).to(cuda))
    grads = []
    for flag in (False, True):
        for m in unet.modules():
            if hasattr(m, "use_checkpoint"):
                m.use_checkpoint = flag
            if hasattr(m, "checkpoint") and isinstance(getattr(m, "checkpoint"), bool):
                m.checkpoint = flag
        unet.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        loss, _ = model.p_losses(*args, **kw)
        loss.backward()
        grads.append({n: p.grad.clone() for n, p in unet.named_parameters()})
    assert all(torch.equal(grads[0][n], grads[1][n]) for n in grads[0])


def test_full_width_unet_gradients_match_autograd_of_the_cpu_oracle(cuda):
    """The REAL topology — 1.44 B parameters, channels 320 / 640 / 1280, 4 levels, 16 spatial + 17 temporal transformers — on a small
    clip (4 frames of 16 x 24 latents), so that autograd of the CPU oracle stays at tens of seconds: p_losses loss and the gradient
    of all 1520-odd parameter tensors (K-sliced weight gradients, every conv / linear width of the production model)."""
    from mudg_amd import configs, factory
    from oracle import unet as o_unet
    model = factory.build_synthetic_model("512", cuda, seed=9).train()
    unet = model.model.diffusion_model
    for m in unet.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    inp = factory.synthetic_inputs(model, "512", 1, cuda, seed=13, latent_shape=(4, 4, 16, 24))
    t = torch.tensor([431], device=cuda)
    noise = rnd(*inp["x_T"].shape, seed=5).to(cuda)
    loss, _ = model.p_losses(inp["x_T"], inp["cond"], t, noise=noise, class_label=inp["class_label"], fs=inp["fs"])
    loss.backward()
    ref_sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in unet.state_dict().items()}
    x0, nz, concat = inp["x_T"].cpu(), noise.cpu(), inp["cond"]["c_concat"][0].cpu()
    sac, s1m = model.sqrt_alphas_cumprod.cpu()[t.cpu()], model.sqrt_one_minus_alphas_cumprod.cpu()[t.cpu()]
    x_noisy, target = sac * x0 + s1m * nz, sac * nz - s1m * x0
    cfg = dict(configs.UNET_MDM)
    pred = o_unet.unet_forward.__wrapped__(ref_sd, cfg, torch.cat([x_noisy, concat], 1), t.cpu(), inp["class_label"][:, 0].cpu(),
                                           inp["cond"]["c_crossattn"][0].cpu(), inp["fs"].cpu())
    want = ((pred - target) ** 2).mean()
    want.backward()
    check("full-width p_losses loss", loss, want.detach(), TOL_NET)
    num = den = 0.0
    worst, n = 0.0, 0
    for name, p in unet.named_parameters():
        gr = ref_sd[name].grad
        assert p.grad is not None and gr is not None, name
        d = (p.grad.float().cpu() - gr).norm().item()
        num += d * d; den += gr.norm().item() ** 2
        worst = max(worst, d / max(gr.norm().item(), 1e-30))
        n += 1
    total = (num / den) ** 0.5
    print(f"[{MODE}] full-width UNet ({sum(p.numel() for p in unet.parameters()) / 1e9:.2f} B parameters, {n} tensors): gradients vs oracle autograd "
          f"overall rel-L2 {total:.3e}, worst single tensor {worst:.3e} (bound {TOL_NET:g} overall)")
    assert total < TOL_NET


def test_gradient_clipping_equals_torch_clip_grad_norm(cuda):
    """The reference's trainer clips the global gradient 2-norm to 0.5 (configs/stage2-1024_mdm_waymo/config.yaml); GradientClipper
    does it on the device: same norm, same scaled gradients as torch.nn.utils.clip_grad_norm_, ragged tensor sizes included."""
    from mudg_amd.train import step
    shapes = [(5,), (300, 7), (16384,), (16385,), (3, 3, 64, 64), (1,)]
    for scale, clipped in ((1.0, True), (1e-4, False)):               # above the bound (scaled down) and below it (untouched)
        mine = [torch.nn.Parameter(torch.zeros(sh, device=cuda)) for sh in shapes]
        ref = [torch.nn.Parameter(torch.zeros(sh, device=cuda)) for sh in shapes]
        for i, (a, b) in enumerate(zip(mine, ref)):
            g = (rnd(*shapes[i], seed=20 + i) * scale).to(cuda)
            a.grad, b.grad = g.clone(), g.clone()
        clip = step.GradientClipper(mine, 0.5)
        stat = clip()
        want = torch.nn.utils.clip_grad_norm_(ref, 0.5)
        assert abs(float(stat[0]) - float(want)) <= 1e-6 * float(want)
        assert (float(stat[1]) < 1.0) == clipped
        for a, b in zip(mine, ref):
            check(f"clipped gradient {tuple(a.shape)}", a.grad, b.grad, 1e-6)
        again = clip()                                                 # the table is reused; the norm is now min(norm, 0.5)
        assert float(again[0]) <= 0.5 * (1 + 1e-5) + 1e-12


def test_adamw_step_matches_torch_and_training_reduces_the_loss(cuda):
    from mudg_amd.train import step
    p = torch.nn.Parameter(rnd(300, 7, seed=1).to(cuda))
    q = torch.nn.Parameter(p.detach().clone())
    mine, ref = step.AdamW([p], lr=1e-2, weight_decay=0.05), torch.optim.AdamW([q], lr=1e-2, weight_decay=0.05)
    for i in range(3):
        gr = rnd(300, 7, seed=10 + i).to(cuda)
        p.grad, q.grad = gr.clone(), gr.clone()
        mine.step(); ref.step()
    check("AdamW 3 steps", p, q, 1e-6)
    model, g, _ = _tiny_model(cuda)
    model.learning_rate = 2e-4
    opt = model.configure_optimizers()
    shp = g["shape"]
    x, ctx = unet_inputs(g["cfg"], shp, g["seed"])
    batch = dict(x_start=x[:, :4].contiguous().to(cuda), cond={"c_crossattn": [ctx.to(cuda)], "c_concat": [x[:, 4:].contiguous().to(cuda)]},
                 t=torch.tensor([700, 420, 100])[:shp["B"]].to(cuda), noise=rnd(shp["B"], 4, shp["T"], shp["H"], shp["W"], seed=3).to(cuda),
                 class_label=torch.tensor([0, 500, 1])[:shp["B"], None].to(cuda), fs=torch.full((shp["B"],), 10).to(cuda))
    losses = []
    for _ in range(6):                                     # the same batch: the loss must go down
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    print(f"[{MODE}] six AdamW steps on one batch: loss {losses[0]:.5f} -> {losses[-1]:.5f}")
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]


def test_resampler_forward_and_every_parameter_gradient_match_autograd_of_the_cpu_oracle(cuda):
    """The image-token Perceiver trains in the MuDG configs (image_proj_model_trainable; ddpm3d.py:1281-1284): in training mode
    its forward runs on the autograd Functions (mudg_amd/train/resampler.py) and a loss on the context tokens must reach every
    one of its parameters — compared with torch.autograd of oracle/resampler.py, which is pinned to the reference's output."""
    from helpers import seeding
    from lvdm.modules.encoders.resampler import Resampler
    from oracle import resampler as o_res
    g = golden("resampler.pt")
    net = Resampler(**g["cfg"])
    sd = seeded_sd(g["param_shapes"], g["seed"], g["checksum"])
    net.load_state_dict(sd, strict=True)
    net = net.to(cuda).train()
    x = seeding.seeded_input("clip_tokens", (3, 257, g["cfg"]["embedding_dim"]), g["seed"])
    xg = x.to(cuda).requires_grad_(True)
    out = net(xg)
    assert out.requires_grad and out.shape == g["out"].shape
    check("resampler training forward vs the reference fixture", out, g["out"], TOL)
    probe = rnd(*out.shape, seed=5)
    (out * probe.to(cuda)).sum().backward()
    ref_sd = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    want = o_res.forward.__wrapped__(ref_sd, xr, g["cfg"]["heads"], g["cfg"]["depth"])
    (want * probe).sum().backward()
    check("resampler d(input tokens)", xg.grad, xr.grad, 4 * TOL)
    num = den = worst = 0.0
    missing = [k for k, p in net.named_parameters() if p.grad is None]
    assert not missing, missing
    for k, p in net.named_parameters():
        gr = ref_sd[k].grad
        d = (p.grad.float().cpu() - gr).norm().item()
        num += d * d; den += gr.norm().item() ** 2
        worst = max(worst, d / max(gr.norm().item(), 1e-30))
    total = (num / den) ** 0.5
    print(f"[{MODE}] Resampler parameter gradients vs oracle autograd: {len(list(net.parameters()))} tensors, overall rel-L2 {total:.3e}, "
          f"worst single tensor {worst:.3e}")
    assert total < 4 * TOL and worst < 20 * TOL
    # evaluation mode / no_grad keeps the fused inference executor (no graph)
    with torch.no_grad():
        assert not net(x.to(cuda)).requires_grad
    assert not net.eval()(x.to(cuda)).requires_grad


def test_latent_diffusion_forward_matches_the_reference_fixture(cuda, monkeypatch):
    """model(x, c, **kw) — the training entry of ddpm3d.py:711-715: random t, dynamic rescale of the latents (scale_arr[t]; the
    MDM configs enable it), then p_losses — against the reference's own call on the tiny model (tests/golden/training_forward.pt,
    made by make_golden.py: timesteps and noise as the reference drew them are replayed here)."""
    from lvdm.models.ddpm3d import LatentVisualDiffusion
    from helpers import seeding
    g = golden("training_forward.pt")
    ident = {"target": "torch.nn.Identity"}
    model = LatentVisualDiffusion(
        img_cond_stage_config=ident, image_proj_stage_config=ident, cond_stage_config=ident, first_stage_config=ident,
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": g["unet_cfg"]}, **g["diffusion_cfg"])
    model.model.diffusion_model.load_state_dict(seeded_sd(g["unet_param_shapes"], g["seed"], g["unet_checksum"]), strict=True)
    model = model.to(cuda).eval()
    shp = g["shape"]
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    x = seeding.seeded_input("x_start", (B, 4, T, H, W), g["input_seed"])
    ctx = seeding.seeded_input("ctx_train", (B, 77 + 16 * T, g["unet_cfg"]["context_dim"]), g["input_seed"])
    concat = seeding.seeded_input("c_concat_train", (B, 8, T, H, W), g["input_seed"], 0.18215 * 5)
    cond = {"c_crossattn": [ctx.to(cuda)], "c_concat": [concat.to(cuda)]}
    # replay the reference's draws: t from randint, the noise as (offset term 0) + (full-size term = the recorded sum)
    monkeypatch.setattr(torch, "randint", lambda *a, **k: g["t"].to(cuda))
    monkeypatch.setattr(torch, "randn", lambda *a, **k: torch.zeros(*a, device=k.get("device")))
    monkeypatch.setattr(torch, "randn_like", lambda t_, **k: g["noise"].to(t_.device))
    seen = {}
    orig = model.q_sample
    monkeypatch.setattr(model, "q_sample", lambda x_start, t, noise=None: (seen.update(x=x_start.clone(), t=t.clone()), orig(x_start, t, noise))[1])
    with torch.no_grad():
        loss, info = model(x.to(cuda), cond, fs=g["fs"].to(cuda), class_label=g["class_label"].to(cuda))
    monkeypatch.undo()
    assert torch.equal(seen["t"].cpu(), g["t"])
    check("dynamic rescale of the latents (scale_arr[t] x)", seen["x"], g["x_rescaled"], 1e-6)
    tol = {"bf16": 3e-2, "fp16": 5e-3, "bf16x3": 3e-4, "bf16x6": 3e-5}[MODE]
    check("forward() loss vs the reference", loss, g["loss"], tol)
    assert set(info) == set(g["loss_dict"])
    for k, v in g["loss_dict"].items():
        assert abs(float(info[k]) - float(v)) < tol * abs(float(v)), (k, float(info[k]), float(v))


def test_inference_after_an_optimizer_step_sees_the_new_weights(cuda):
    """mudg AdamW writes parameters through raw pointers; the inference side recognises changed weights by (data_ptr, _version) —
    packed operand copies, captured graphs, cached K / V^T.  After a step an evaluation forward must equal that of a freshly
    built model holding the updated weights, and differ from the one before the step."""
    model, g, _ = _tiny_model(cuda)
    model.learning_rate = 5e-3
    opt = model.configure_optimizers()
    shp = g["shape"]
    x, ctx = unet_inputs(g["cfg"], shp, g["seed"])
    t = torch.tensor([700, 420, 100])[:shp["B"]].to(cuda)
    kw = dict(c_label=torch.tensor([0, 500, 1])[:shp["B"]].to(cuda), context=ctx.to(cuda), fs=torch.full((shp["B"],), 10).to(cuda))
    unet = model.model.diffusion_model

    def infer(net):
        net.eval()
        with torch.no_grad():
            y = net(x.to(cuda), t, **kw).clone()
        net.train()
        return y
    before = infer(unet)
    versions = [p._version for p in unet.parameters()]
    batch = dict(x_start=x[:, :4].contiguous().to(cuda), cond={"c_crossattn": [ctx.to(cuda)], "c_concat": [x[:, 4:].contiguous().to(cuda)]},
                 t=t, noise=rnd(shp["B"], 4, shp["T"], shp["H"], shp["W"], seed=3).to(cuda),
                 class_label=kw["c_label"][:, None], fs=kw["fs"])
    opt.zero_grad(set_to_none=True)
    model.training_step(batch).backward()
    opt.step()
    assert all(p._version > v for p, v in zip(unet.parameters(), versions) if p.grad is not None)
    after = infer(unet)
    fresh = copy.deepcopy(unet)                            # new module objects: nothing cached can be reused
    for m in fresh.modules():
        for k in [k for k in vars(m) if k.startswith("_mudg")]:
            delattr(m, k)
    want = infer(fresh)
    assert torch.equal(after, want), rel_l2(after, want)
    assert rel_l2(after, before) > 1e-4


def test_multi_tensor_adamw_equals_torch_over_many_tensors(cuda):
    from mudg_amd import hip
    from mudg_amd.train import step
    chunk = hip.lib().mudg_clip_chunk()
    shapes = [(7,), (chunk + 13,), (300, 7), (2 * chunk,), (1,), (3, chunk // 2 + 5)]
    ps = [torch.nn.Parameter(rnd(*s, seed=i).to(cuda)) for i, s in enumerate(shapes)]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mine, ref = step.AdamW(ps, lr=3e-3, weight_decay=0.02), torch.optim.AdamW(qs, lr=3e-3, weight_decay=0.02)
    for it in range(3):
        for i, (p, q) in enumerate(zip(ps, qs)):
            gr = rnd(*p.shape, seed=100 * it + i).to(cuda)
            p.grad, q.grad = (gr.clone(), gr.clone()) if not (it == 1 and i == 4) else (None, None)      # a tensor that skips a step
        mine.step(); ref.step()
    for i, (p, q) in enumerate(zip(ps, qs)):
        check(f"multi-tensor AdamW tensor {i} {tuple(p.shape)}", p, q, 1e-6)


def test_adamw_load_state_dict_replaces_the_moments_the_kernel_reads(cuda):
    """The multi-tensor launch reads a cached table of raw addresses, the moments' included: optimizer.load_state_dict (which
    replaces the moment tensors) must rebuild it, or the kernel would keep updating the freed buffers and ignore the restored ones."""
    from mudg_amd.train import step
    shapes = [(33,), (128, 9), (5,)]
    ps = [torch.nn.Parameter(rnd(*s, seed=i).to(cuda)) for i, s in enumerate(shapes)]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mine, ref = step.AdamW(ps, lr=2e-3, weight_decay=0.01), torch.optim.AdamW(qs, lr=2e-3, weight_decay=0.01)

    def one(it):
        for i, (p, q) in enumerate(zip(ps, qs)):
            gr = rnd(*p.shape, seed=50 * it + i).to(cuda)
            if p.grad is None:
                p.grad, q.grad = gr.clone(), gr.clone()
            else:                                           # in place: the parameter / gradient addresses (the old cache key) stay the same
                p.grad.copy_(gr); q.grad.copy_(gr)
        mine.step(); ref.step()
    one(0); one(1)
    saved_m, saved_r = copy.deepcopy(mine.state_dict()), copy.deepcopy(ref.state_dict())
    saved_p = [p.detach().clone() for p in ps]
    one(2); one(3)
    with torch.no_grad():
        for p, q, s in zip(ps, qs, saved_p):
            p.copy_(s); q.copy_(s)
    mine.load_state_dict(saved_m); ref.load_state_dict(saved_r)
    one(2); one(3)
    for i, (p, q) in enumerate(zip(ps, qs)):
        check(f"AdamW after load_state_dict, tensor {i}", p, q, 1e-6)
