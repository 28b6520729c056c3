"""The tensor-level part of MuDG's inference driver (reference: virtual_render/virtual_pose_render.py —
get_latent_z 54-59, image_guided_synthesis 62-147), with the same argument order, cond / uc dict layout and return
value, on the MI355X path.  The reference file also holds the Waymo frame loaders, PNG/NPY writers and the CLI; those
are host I/O outside the denoising path (SURVEY.md §8(f) rank 3 covers this call sequence, the sliding-window loop with its
autoregressive colour re-feed — `synthesize_windows` below, fed with tensors by the caller — and the per-modality
post-processing in `virtual_render/eval_tools.py`).

Conditioning encoders: `model.embedder` (CLIP image tower) and `model.cond_stage_model` (CLIP text tower) are whatever
modules the config instantiated — they are third-party and not part of this path; `model.image_proj_model` (the
Resampler) and both VAE encodes run on the HIP kernels.
"""
import torch

from lvdm.models.samplers.ddim import DDIMSampler
from lvdm.models.samplers.ddim_multiplecond import DDIMSampler as DDIMSampler_multicond


def get_latent_z(model, videos):
    """(b, c, t, h, w) pixels -> (b, 4, t, h/8, w/8) scaled latents, frame-wise through the VAE encoder."""
    b, c, t, h, w = videos.shape
    frames = videos.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    z = model.encode_first_stage(frames)
    return z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)


def image_guided_synthesis(model, prompts, sparse_x, sparse_depth, class_label, noise_shape, n_samples=1, ddim_steps=50,
                           ddim_eta=1., unconditional_guidance_scale=1.0, cfg_img=None, fs=None, text_input=False,
                           multiple_cond_cfg=False, timestep_spacing="uniform", guidance_rescale=0.0, **kwargs):
    sampler = DDIMSampler(model) if not multiple_cond_cfg else DDIMSampler_multicond(model)
    batch_size = sparse_x.shape[0]
    fs = torch.tensor([fs] * batch_size, dtype=torch.long, device=model.device)
    if not text_input:
        prompts = [""] * batch_size

    img = sparse_x[:, :, 0]                                   # conditioning frame, (b, c, h, w)
    img_emb = model.image_proj_model(model.embedder(img))     # (b, 16 t, d)
    cond_emb = model.get_learned_conditioning(prompts)        # (b, 77, d)
    cond = {"c_crossattn": [torch.cat([cond_emb, img_emb], dim=1)]}
    hybrid = model.model.conditioning_key == "hybrid"
    if hybrid:
        sparse_z = get_latent_z(model, sparse_x)
        sparse_depth_z = get_latent_z(model, sparse_depth)
        kwargs.update({"sparse_x": sparse_z, "class_label": class_label})
        img_cat_cond = torch.cat([sparse_z, sparse_depth_z], dim=1)
        cond["c_concat"] = [img_cat_cond]

    uc = None
    if unconditional_guidance_scale != 1.0:
        if model.uncond_type == "empty_seq":
            uc_emb = model.get_learned_conditioning(batch_size * [""])
        else:
            uc_emb = torch.zeros_like(cond_emb)
        uc_img_emb = model.image_proj_model(model.embedder(torch.zeros_like(img)))
        uc = {"c_crossattn": [torch.cat([uc_emb, uc_img_emb], dim=1)]}
        if hybrid:
            uc["c_concat"] = [img_cat_cond]
    if multiple_cond_cfg and cfg_img != 1.0:
        uc_2 = {"c_crossattn": [torch.cat([uc_emb, img_emb], dim=1)]}
        if hybrid:
            uc_2["c_concat"] = [img_cat_cond]
        kwargs.update({"unconditional_conditioning_img_nonetext": uc_2})
    else:
        kwargs.update({"unconditional_conditioning_img_nonetext": None})

    variants = []
    for _ in range(n_samples):
        samples, _ = sampler.sample(S=ddim_steps, conditioning=cond, batch_size=batch_size, shape=noise_shape[1:],
                                    verbose=False, unconditional_guidance_scale=unconditional_guidance_scale,
                                    unconditional_conditioning=uc, eta=ddim_eta, cfg_img=cfg_img, mask=None, x0=None,
                                    fs=fs, timestep_spacing=timestep_spacing, guidance_rescale=guidance_rescale,
                                    **kwargs)
        variants.append(model.decode_first_stage(samples))
    return torch.stack(variants).permute(1, 0, 2, 3, 4, 5)    # (batch, variants, c, t, h, w)


def synthesize_windows(model, windows, noise_shape, video_length=16, **synthesis_kwargs):
    """The sliding-window loop of run_inference_multi (virtual_pose_render.py:222-355) on tensors.

    `windows` yields, per window, a dict with the three modality streams in the reference's order (colour, depth,
    semantic): "sparse" (3, c, t, h, w), "dense" (3, c, t, h, w), "sparse_depth" (3, c, t, h, w), "class_label" (3, 1) —
    what get_color_frames / get_depth_frames / get_semantic_frames / get_sparse_depth load from disk there.  Consecutive
    windows overlap by half (the index advances by video_length // 2, :247); before a window is synthesised, the colour
    stream's first video_length // 2 sparse frames are replaced by the last video_length // 2 frames generated for the
    previous window (:269-274) and its frame 0 by the dense frame 0 (:275).  Returns the list of clamped samples
    (3, n_samples, c, t, h, w) per window, as batch_samples after :243."""
    half = video_length // 2
    carry = None
    results = []
    for win in windows:
        sparse = win["sparse"].clone()
        if carry is not None:
            sparse[0, :, 0:half] = carry[:, 0:half]
            sparse[0, :, 0] = win["dense"][0, :, 0]
        dev = model.device
        samples = image_guided_synthesis(model, [""] * sparse.shape[0], sparse.to(dev), win["sparse_depth"].to(dev),
                                         win["class_label"].to(dev), noise_shape, **synthesis_kwargs)
        samples = torch.clamp(samples.float(), -1., 1.)
        results.append(samples)
        for nn in range(samples.shape[0]):
            if int(win["class_label"][nn, 0]) == 0:                        # the colour stream re-feeds itself
                carry = samples[nn, 0, :, half:video_length].to(sparse.device)      # (c, half, h, w)
    return results
