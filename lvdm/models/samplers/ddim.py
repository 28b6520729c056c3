"""DDIM sampler at the drop-in boundary (reference: lvdm/models/samplers/ddim.py, DDIMSampler 10-317).

Same constructor, `make_schedule`, `sample`, `ddim_sampling`, `p_sample_ddim`, `decode` and `stochastic_encode` signatures
and return values.  What differs is where the arithmetic runs: the schedule is derived once on the host (same
float64/float32 mix as the reference, see utils_diffusion), each step's scalars are kept as host floats instead of
1-element device tensors, and the whole per-step update — classifier-free guidance, guidance rescale (per-sample std),
v or eps -> (eps, x0), dynamic rescale and the x_{t-1} formula — is ONE fused HIP launch (mudg_ddim_step) after the UNet
passes.  The options MuDG's own drivers leave at their defaults are implemented too, around the same launch: mask
blending against a (noised) original latent, a `timesteps` prefix of the DDIM schedule, the full-schedule "original
steps" mode, temperature, noise dropout, eps-parameterised models with an optional score corrector, and x0 quantisation
for first stages that have a `quantize` method (AutoencoderKL has none — there, as in the reference, it is an
AttributeError).
"""
import numpy as np
import torch

from lvdm.common import noise_like
from lvdm.models.utils_diffusion import make_ddim_sampling_parameters, make_ddim_timesteps, rescale_noise_cfg


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.counter = 0
        # Run the conditional and unconditional UNet passes of classifier-free guidance as ONE pass over a doubled
        # batch (per-sample arithmetic is unchanged: every op of the UNet is per clip / per frame / per pixel).  The
        # reference runs them back to back (ddim.py:221-222); batching halves the launch count and fills the chip at
        # the coarse levels (M = 2304 rows at ds8).  Set to False to reproduce the reference's call sequence.
        self.batch_cfg = True
        # ... and, when those passes differ only in their cross-attention tokens, let the UNet run the part of the network
        # that comes before the first cross-attention once for all of them (see _batched_passes).
        self.share_guidance_prefix = True

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    # ------------------------------------------------------------------ schedule (host)
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps,
                                                  verbose=verbose)
        ac = self.model.alphas_cumprod.detach().float().cpu()
        assert ac.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        if self.model.use_dynamic_rescale:
            sa = self.model.scale_arr.detach().float().cpu()[self.ddim_timesteps]
            self.ddim_scale_arr = sa
            self.ddim_scale_arr_prev = torch.cat([sa[0:1], sa[:-1]])
        prev = self.model.alphas_cumprod_prev.detach().float().cpu()
        self.register_buffer("alphas_cumprod", ac)
        self.register_buffer("alphas_cumprod_prev", prev)
        self.register_buffer("sqrt_alphas_cumprod", self.model.sqrt_alphas_cumprod.detach().float().cpu())
        self.register_buffer("sqrt_one_minus_alphas_cumprod",
                             self.model.sqrt_one_minus_alphas_cumprod.detach().float().cpu())
        sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(ac, self.ddim_timesteps, ddim_eta, verbose=verbose)
        self.register_buffer("ddim_sigmas", sigmas)
        self.register_buffer("ddim_alphas", alphas)
        self.register_buffer("ddim_alphas_prev", alphas_prev)
        self.register_buffer("ddim_sqrt_one_minus_alphas", torch.sqrt(1. - alphas))
        # the full-schedule ("original steps") walk has its own sigma table (ddim.py:55-58)
        self.register_buffer("ddim_sigmas_for_original_num_steps",
                             ddim_eta * torch.sqrt((1 - prev) / (1 - ac) * (1 - ac / prev)))

    def step_coefficients(self, index, cfg_scale, guidance_rescale, temperature=1.0, use_original_steps=False):
        """The host scalars of mudg_ddim_step for schedule position `index`, each rounded to fp32 where the reference
        rounds it (torch.full((b,1,1,1,1), value) of a python/np/tensor scalar, ddim.py:251-254,262-266).
        use_original_steps: `index` is a DDPM timestep and the full-schedule tables apply (ddim.py:241-246)."""
        f32 = lambda v: torch.full((1,), float(v), dtype=torch.float32)
        if use_original_steps:
            t = int(index)
            a_t, a_prev = f32(self.alphas_cumprod[t]), f32(self.alphas_cumprod_prev[t])
            sqrt_1m_at, sigma = f32(self.sqrt_one_minus_alphas_cumprod[t]), f32(self.ddim_sigmas_for_original_num_steps[t])
        else:
            t = int(self.ddim_timesteps[index])
            a_t, a_prev = f32(self.ddim_alphas[index]), f32(self.ddim_alphas_prev[index])
            sqrt_1m_at, sigma = f32(self.ddim_sqrt_one_minus_alphas[index]), f32(self.ddim_sigmas[index])
        rescale = torch.ones(1)
        if self.model.use_dynamic_rescale:
            rescale = f32(self.ddim_scale_arr_prev[index]) / f32(self.ddim_scale_arr[index])
        dir_coef = (1. - a_prev - sigma ** 2).sqrt()
        if self.model.parameterization == "v":         # x0 / eps from the model's own tables at timestep t (ddpm3d.py:239-251)
            ca, cb, eps_form = self.sqrt_alphas_cumprod[t], self.sqrt_one_minus_alphas_cumprod[t], 0.0
        else:                                          # eps-prediction: x0 = (x - sqrt(1 - a_t) e) / sqrt(a_t)  (ddim.py:257-258)
            ca, cb, eps_form = a_t.sqrt(), sqrt_1m_at, 1.0
        return [float(cfg_scale), float(guidance_rescale), float(ca), float(cb), float(rescale), float(a_prev.sqrt()),
                float(dir_coef), float(sigma * temperature), 0.0, eps_form]

    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, schedule_verbose=False, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, precision=None, fs=None,
               timestep_spacing="uniform", guidance_rescale=0.0, **kwargs):
        if conditioning is not None:
            first = conditioning[list(conditioning.keys())[0]] if isinstance(conditioning, dict) else conditioning
            cbs = (first[0] if isinstance(first, (list, tuple)) else first).shape[0]
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_discretize=timestep_spacing, ddim_eta=eta, verbose=schedule_verbose)
        if len(shape) not in (3, 4):
            raise ValueError(f"shape must be (C, H, W) or (C, T, H, W), got {shape}")
        size = (batch_size, *shape)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, verbose=verbose,
                                  precision=precision, fs=fs, guidance_rescale=guidance_rescale, **kwargs)

    def _walk(self, timesteps, original):
        """(timestep values in sampling order, their count) for the three ways the reference picks the walk
        (ddim.py:152-160): the whole DDIM schedule, a prefix of it (`timesteps` = how many of its entries, minus one as
        the reference counts), or t = timesteps-1 .. 0 of the DDPM schedule."""
        if original:
            count = self.ddpm_num_timesteps if timesteps is None else int(timesteps)
            return list(range(count - 1, -1, -1)), count
        steps = self.ddim_timesteps
        if timesteps is not None:
            full = steps.shape[0]
            steps = steps[:int(min(timesteps / full, 1) * full) - 1]
        return [int(v) for v in np.flip(steps)], int(steps.shape[0])

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, verbose=True, precision=None, fs=None, guidance_rescale=0.0,
                      **kwargs):
        clean_cond = kwargs.pop("clean_cond", False)
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        walk, total = self._walk(timesteps, ddim_use_original_steps)
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        if mask is not None:
            assert x0 is not None, "mask blending needs the original latent x0"
            mask = mask.to(device=device, dtype=torch.float32)
            x0 = x0.to(device=device, dtype=torch.float32)
        iterator = walk
        if verbose:
            from tqdm import tqdm
            iterator = tqdm(walk, desc="DDIM Sampler", total=total)
        for i, step in enumerate(iterator):
            index = total - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if mask is not None:        # keep the masked region on the original's trajectory (ddim.py:173-180)
                keep = x0 if clean_cond else self.model.q_sample(x0, ts)
                img = keep * mask + (1. - mask) * img
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, use_original_steps=ddim_use_original_steps,
                                              quantize_denoised=quantize_denoised,
                                              temperature=temperature, noise_dropout=noise_dropout,
                                              score_corrector=score_corrector, corrector_kwargs=corrector_kwargs,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning, mask=mask, x0=x0,
                                              fs=fs, guidance_rescale=guidance_rescale, **kwargs)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        self.clear_conditioning_cache()          # the run is over: do not pin its K / V^T buffers and cond dicts on the sampler
        return img, intermediates

    # ------------------------------------------------------------------ one step
    def _model_outputs(self, x, t, c, unconditional_conditioning, unconditional_guidance_scale, kwargs):
        """The UNet passes of one step: (e_cond, e_uncond or None, e_image_only or None, cfg_img coefficient).  Two-way
        guidance here (ddim.py:213-225); ddim_multiplecond overrides it with the three-way form."""
        guided = unconditional_conditioning is not None and unconditional_guidance_scale != 1.
        if not guided:
            return self.model.apply_model(x, t, c, **kwargs), None, None, 0.0
        if not isinstance(c, (torch.Tensor, dict)):
            raise NotImplementedError("guided sampling needs a tensor or dict conditioning")     # as ddim.py:223-224
        pair = self._batched_cfg(x, t, c, unconditional_conditioning, kwargs) if self.batch_cfg else None
        if pair is None:
            pair = (self.model.apply_model(x, t, c, **kwargs),
                    self.model.apply_model(x, t, unconditional_conditioning, **kwargs))
        return pair[0], pair[1], None, 0.0

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, uc_type=None,
                      conditional_guidance_scale_temporal=None, mask=None, x0=None, guidance_rescale=0.0, **kwargs):
        from mudg_amd import ops
        x = x.float().contiguous()
        e_c, e_u, e_m, cfg_img = self._model_outputs(x, t, c, unconditional_conditioning, unconditional_guidance_scale,
                                                     kwargs)
        guided = e_u is not None
        coef = self.step_coefficients(index, unconditional_guidance_scale if guided else 1.0,
                                      guidance_rescale if guided else 0.0, temperature, use_original_steps)
        coef[8] = float(cfg_img)
        if score_corrector is not None:
            # the corrector edits eps AFTER guidance (ddim.py:233-235), so the guided output is formed here and the fused
            # update runs unguided on the corrected eps
            assert self.model.parameterization == "eps", "not implemented"
            e_c = self._guided_output(e_c, e_u, e_m, coef)
            e_c = score_corrector.modify_score(self.model, e_c, x, t, c, **(corrector_kwargs or {}))
            e_u = e_m = None
            coef[0], coef[1] = 1.0, 0.0
        # always drawn, as the reference does (ddim.py:272: sigma_t * noise_like(...)): with eta = 0 the term is 0 * noise, but
        # the device generator advances identically, so later draws under the same seed (n_samples > 1) match
        noise = noise_like(x.shape, x.device, repeat_noise)
        if noise_dropout > 0.:
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        x_prev, pred_x0 = ops.ddim_step(x, e_c, e_u, noise, coef, e_m=e_m)
        if quantize_denoised:           # first stages with a codebook only; x_{t-1} follows the quantised x0 (ddim.py:267-275)
            snapped, _, *_ = self.model.first_stage_model.quantize(pred_x0)
            x_prev = x_prev + coef[5] * (snapped - pred_x0)
            pred_x0 = snapped
        return x_prev, pred_x0

    @staticmethod
    def _guided_output(e_c, e_u, e_m, coef):
        if e_u is None:
            return e_c
        s, phi, s_img = coef[0], coef[1], coef[8]
        out = e_u + s * (e_c - e_u) if e_m is None else e_u + s_img * (e_m - e_u) + s * (e_c - e_m)
        return rescale_noise_cfg(out, e_c, phi) if phi > 0.0 else out

    def _batched_cfg(self, x, t, c, uc, kwargs):
        """[cond | uncond] in one apply_model call; None when the conditionings cannot be stacked."""
        return self._batched_passes(x, t, [c, uc], kwargs)

    def _batched_passes(self, x, t, conds, kwargs):
        """One apply_model call for the len(conds) guidance passes of a step; None if the dicts cannot be stacked.
        The stacked conditioning is the same at every step of a run, so it is built once and kept while the caller keeps
        passing the same tensors: the cross-attention tokens stacked along the batch AND made ready for the UNet
        (UNetModel.prepare_context: operand rows + every cross-attention layer's K / V^T projections, which the reference
        recomputes at each of the 50 steps although the tokens never change).
        When the passes differ ONLY in those tokens — MuDG's driver hands the same c_concat latents to all of them — the
        latents are passed once (batch b) with the n b stacked contexts: the UNet then runs its context-free prefix (stem,
        init_attn, first ResBlock, first full-resolution self-attention) once instead of n times.  Otherwise x, t and the
        per-sample keyword tensors ride along n times."""
        first = conds[0]
        if not all(isinstance(cd, dict) and set(cd) == set(first) for cd in conds):
            return None
        b, n = x.shape[0], len(conds)
        merged, shared = self._merged_conditioning(conds, x.shape[2] if x.dim() == 5 else None)
        if merged is None:
            return None
        if shared:
            out = self.model.apply_model(x, t, merged, **kwargs)
        else:
            kw = {}
            for k, v in kwargs.items():          # per-sample tensors ride along n times; everything else is shared
                kw[k] = torch.cat([v] * n, 0) if (torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == b) else v
            out = self.model.apply_model(torch.cat([x] * n, 0), torch.cat([t] * n, 0), merged, **kw)
        return tuple(out[i * b:(i + 1) * b] for i in range(n))

    def _merged_conditioning(self, conds, frames):
        """(stacked conditioning dict, shared-latents flag) — see _batched_passes; (None, False) if not stackable."""
        def ident(v):
            try:
                ver = v._version
            except RuntimeError:
                ver = 0
            return (v.data_ptr(), ver, tuple(v.shape), v.dtype)

        def same(u, v):
            return u is v or (u.data_ptr() == v.data_ptr() and u.shape == v.shape and u.stride() == v.stride() and u.dtype == v.dtype)

        first = conds[0]
        for key in first:
            lists = [cd[key] for cd in conds]
            if not all(isinstance(l, (list, tuple)) and len(l) == len(lists[0]) for l in lists):
                return None, False
            if any((not torch.is_tensor(v)) or v.shape != lists[0][i].shape for l in lists for i, v in enumerate(l)):
                return None, False
        unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
        # the key names the tensors (address, version, shape, dtype), the frame count the prepared context was laid out for
        # and the UNet whose projections it holds; inference-mode tensors do not track versions, so an in-place update of
        # one would go unseen — such conditionings are stacked afresh at every call instead of being cached
        cacheable = not any(v.is_inference() for cd in conds for key in cd for v in cd[key])
        sig = (bool(self.share_guidance_prefix), frames, id(unet)) + tuple((key, tuple(ident(v) for v in cd[key])) for cd in conds for key in sorted(cd))
        cache = self.__dict__.get("_merged_cache")
        if cacheable and cache is not None and cache[0] == sig:
            return cache[1], cache[3]
        prepare = frames is not None and "c_crossattn" in first and hasattr(unet, "prepare_context") \
            and all(v.is_cuda for cd in conds for v in cd["c_crossattn"])
        # guidance replicas: every input but the tokens is literally the same tensor in all passes
        shared = prepare and self.share_guidance_prefix and all(
            same(cd[key][i], first[key][i]) for cd in conds for key in first if key != "c_crossattn"
            for i in range(len(first[key])))
        merged = {}
        for key in first:
            if shared and key != "c_crossattn":
                merged[key] = list(first[key])
            else:
                merged[key] = [torch.cat([cd[key][i] for cd in conds], 0) for i in range(len(first[key]))]
        if prepare:
            tokens = merged["c_crossattn"][0] if len(merged["c_crossattn"]) == 1 else torch.cat(merged["c_crossattn"], 1)
            merged["c_crossattn"] = [unet.prepare_context(tokens, frames)]
        # the source dicts are kept alive while cached (their addresses are the key); ONE entry, replaced by the next run's
        # and dropped by clear_conditioning_cache() (sample() / decode() call it when they return)
        self._merged_cache = (sig, merged, [cd for cd in conds], shared) if cacheable else None
        return merged, shared

    def clear_conditioning_cache(self):
        """Drop the stacked / projected conditioning of the last run (per-layer K / V^T buffers and the caller's cond dicts)."""
        self._merged_cache = None

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        """Denoise a latent from schedule position t_start down to 0 (ddim.py:281-301): the counterpart of
        stochastic_encode for image-to-image style use."""
        steps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        steps = steps[:t_start]
        total = int(steps.shape[0])
        print(f"Running DDIM Sampling with {total} timesteps")
        x_dec = x_latent
        for i, step in enumerate(np.flip(steps)):
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.p_sample_ddim(x_dec, cond, ts, index=total - i - 1, use_original_steps=use_original_steps,
                                          unconditional_guidance_scale=unconditional_guidance_scale,
                                          unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
        self.clear_conditioning_cache()
        return x_dec

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """q(x_t | x_0) at schedule position(s) `t` (a long tensor, one entry per sample) of the DDIM (or full) schedule
        (ddim.py:303-317).  One HIP launch (the per-sample linear combination the v-parameterisation helpers use)."""
        from mudg_amd import ops
        if use_original_steps:
            ca, cb = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod
        else:
            ca, cb = torch.sqrt(self.ddim_alphas), self.ddim_sqrt_one_minus_alphas
        if noise is None:
            noise = torch.randn_like(x0)
        pick = lambda tab: torch.as_tensor(tab, dtype=torch.float32).to(x0.device)[t]
        return ops.lincomb(x0, noise, pick(ca), pick(cb))
