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


class GuidanceInputs:
    """Everything the guided sampler needs for one batch of modality streams, built once: the conditional dict, the
    unconditional dict (empty prompt + tokens of an all-zero image) and — for three-way guidance — the image-only dict
    (empty prompt + the real image tokens).  All of them share the channel-concat conditioning (the VAE latents of the
    sparse colour and sparse depth renderings); the dict layout is the one LatentDiffusion.apply_model / DiffusionWrapper
    consume ({"c_crossattn": [tokens], "c_concat": [latents]}, virtual_pose_render.py:75-112 in the reference)."""

    def __init__(self, model, prompts, sparse_x, sparse_depth, guided, three_way):
        self.batch = sparse_x.shape[0]
        key_frame = sparse_x[:, :, 0]                                      # the conditioning image of every stream
        tokens = lambda image: model.image_proj_model(model.embedder(image))          # CLIP tower -> Resampler, (b, 16 t, d)
        text = model.get_learned_conditioning(prompts)                     # (b, 77, d)
        image_tokens = tokens(key_frame)
        self.hybrid = model.model.conditioning_key == "hybrid"
        self.concat = None
        self.sparse_z = None
        if self.hybrid:
            self.sparse_z = get_latent_z(model, sparse_x)
            self.concat = torch.cat([self.sparse_z, get_latent_z(model, sparse_depth)], dim=1)
        self.cond = self._entry(text, image_tokens)
        self.uncond = self.image_only = None
        if guided:
            blank = model.get_learned_conditioning(self.batch * [""]) if model.uncond_type == "empty_seq" else torch.zeros_like(text)
            self.uncond = self._entry(blank, tokens(torch.zeros_like(key_frame)))
            if three_way:
                self.image_only = self._entry(blank, image_tokens)

    def _entry(self, text, image_tokens):
        entry = {"c_crossattn": [torch.cat([text, image_tokens], dim=1)]}
        if self.hybrid:
            entry["c_concat"] = [self.concat]
        return entry


def image_guided_synthesis(model, prompts, sparse_x, sparse_depth, class_label, noise_shape, n_samples=1, ddim_steps=50,
                           ddim_eta=1., unconditional_guidance_scale=1.0, cfg_img=None, fs=None, text_input=False,
                           multiple_cond_cfg=False, timestep_spacing="uniform", guidance_rescale=0.0, **kwargs):
    """Same signature and return value as the reference's function (virtual_pose_render.py:62-147): guided DDIM sampling
    of `n_samples` variants for a batch of modality streams, decoded to pixels, (batch, variants, c, t, h, w)."""
    batch = sparse_x.shape[0]
    guided = unconditional_guidance_scale != 1.0
    three_way = bool(multiple_cond_cfg) and cfg_img != 1.0
    inputs = GuidanceInputs(model, prompts if text_input else [""] * batch, sparse_x, sparse_depth, guided, three_way)
    sampler = (DDIMSampler_multicond if multiple_cond_cfg else DDIMSampler)(model)
    extra = dict(kwargs, unconditional_conditioning_img_nonetext=inputs.image_only)
    if inputs.hybrid:
        extra.update(sparse_x=inputs.sparse_z, class_label=class_label)
    frame_rate = torch.tensor([fs] * batch, dtype=torch.long, device=model.device)
    variants = []
    for _ in range(n_samples):
        latents, _ = sampler.sample(S=ddim_steps, conditioning=inputs.cond, batch_size=batch, shape=noise_shape[1:],
                                    verbose=False, unconditional_guidance_scale=unconditional_guidance_scale,
                                    unconditional_conditioning=inputs.uncond, eta=ddim_eta, cfg_img=cfg_img, mask=None, x0=None,
                                    fs=frame_rate, timestep_spacing=timestep_spacing, guidance_rescale=guidance_rescale, **extra)
        variants.append(model.decode_first_stage(latents))
    return torch.stack(variants).permute(1, 0, 2, 3, 4, 5)


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
