"""Latent video diffusion model at the drop-in boundary (reference: lvdm/models/ddpm3d.py).

What the denoising path needs from the reference's Lightning classes, without Lightning: the schedule buffers
(DDPM.register_schedule 123-186, scale_arr 522-527), the conditioning plumbing (LatentDiffusion.apply_model 723-739,
DiffusionWrapper.forward 1309-1324), the v-parameterisation helpers (239-251), first-stage decode (646-671) and the
constructor surface of LatentVisualDiffusion (1033-1055) so MuDG's YAML configs instantiate it unchanged and its
checkpoints load with the same key prefixes (model.diffusion_model.*, first_stage_model.*, image_proj_model.*).
The training step (p_losses, configure_optimizers, training_step) is delegated to mudg_amd.train; the
Lightning loop, logging and the data pipeline around it are not built.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from lvdm.basics import disabled_train
from lvdm.common import default, extract_into_tensor
from lvdm.models.utils_diffusion import make_beta_schedule, rescale_zero_terminal_snr
from utils.utils import instantiate_from_config


def _cfg_get(cfg, key, fallback=None):
    """Configs arrive as OmegaConf nodes, dicts or attr-dicts."""
    if cfg is None:
        return fallback
    if isinstance(cfg, dict) or hasattr(cfg, "keys"):
        try:
            return cfg[key]
        except (KeyError, TypeError):
            return fallback
    return getattr(cfg, key, fallback)


class DiffusionWrapper(nn.Module):
    """Routes conditioning into the UNet.  'hybrid' (MuDG): latent channels = [x | c_concat...] and context =
    cat(c_crossattn).  The channel concat is not materialised: the pieces go to the UNet as a list and are written
    side by side by the layout kernel."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_label=None, c_concat: list = None, c_crossattn: list = None, c_adm=None, s=None,
                mask=None, **kwargs):
        key = self.conditioning_key
        ctx = None
        if c_crossattn is not None:
            ctx = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(list(c_crossattn), 1)
        if key is None:
            return self.diffusion_model(x, t)
        if key == "concat":
            return self.diffusion_model([x] + list(c_concat), t, **kwargs)
        if key == "crossattn":
            return self.diffusion_model(x, t, context=ctx, **kwargs)
        if key == "hybrid":
            return self.diffusion_model([x] + list(c_concat), t, c_label=c_label, context=ctx, **kwargs)
        raise NotImplementedError(f"conditioning_key '{key}' is not on the MuDG path")


class DDPM(nn.Module):
    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None,
                 ignore_keys=[], load_only_unet=False, monitor=None, use_ema=True, first_stage_key="image",
                 image_size=256, channels=3, log_every_t=100, clip_denoised=True, linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, given_betas=None, original_elbo_weight=0., v_posterior=0., l_simple_weight=1.,
                 conditioning_key=None, parameterization="eps", scheduler_config=None, use_positional_encodings=False,
                 learn_logvar=False, logvar_init=0., rescale_betas_zero_snr=False):
        super().__init__()
        assert parameterization in ["eps", "x0", "v"], 'currently only supporting "eps" and "x0" and "v"'
        if use_ema:
            raise NotImplementedError("EMA weights (use_ema) are disabled in every MuDG config and not implemented")
        self.parameterization = parameterization
        self.cond_stage_model = None
        self.clip_denoised, self.log_every_t = clip_denoised, log_every_t
        self.first_stage_key, self.channels = first_stage_key, channels
        self.temporal_length = _cfg_get(_cfg_get(unet_config, "params"), "temporal_length")
        self.image_size = [image_size, image_size] if isinstance(image_size, int) else image_size
        self.use_positional_encodings = use_positional_encodings
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema = False
        self.rescale_betas_zero_snr = rescale_betas_zero_snr
        self.v_posterior, self.original_elbo_weight, self.l_simple_weight = v_posterior, original_elbo_weight, l_simple_weight
        if monitor is not None:
            self.monitor = monitor
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        self.given_betas, self.beta_schedule, self.timesteps, self.cosine_s = given_betas, beta_schedule, timesteps, cosine_s
        self.loss_type = loss_type
        if loss_type != "l2":
            raise NotImplementedError("only the l2 loss of the MuDG configs is implemented")
        # ddpm3d.py:118-121,173-186: per-timestep log-variance (a constant unless learn_logvar) and the vlb weights (ones for v)
        self.learn_logvar = learn_logvar
        if learn_logvar:
            raise NotImplementedError("learn_logvar is off in every MuDG config and not implemented")
        with torch.device("cpu"):       # a plain attribute, as in the reference: real memory even under torch.device("meta")
            self.logvar = torch.full(fill_value=float(logvar_init), size=(self.num_timesteps,))
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys, only_model=load_only_unet)

    @property
    def device(self):
        return self.betas.device

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        with torch.device("cpu"):       # host math even when the model is being built under torch.device("meta")
            self._register_schedule(given_betas, beta_schedule, timesteps, linear_start, linear_end, cosine_s)

    def _register_schedule(self, given_betas, beta_schedule, timesteps, linear_start, linear_end, cosine_s):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        if self.rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        abar = np.cumprod(1. - betas, axis=0)
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        f32 = partial(torch.tensor, dtype=torch.float32)
        for name, val in (("betas", betas), ("alphas_cumprod", abar),
                          ("alphas_cumprod_prev", np.append(1., abar[:-1])),
                          ("sqrt_alphas_cumprod", np.sqrt(abar)),
                          ("sqrt_one_minus_alphas_cumprod", np.sqrt(1. - abar)),
                          ("log_one_minus_alphas_cumprod", np.log(np.maximum(1. - abar, 1e-300)))):
            self.register_buffer(name, f32(val))
        # kept for state_dict compatibility with the reference's checkpoints (not used by v-prediction sampling)
        prev = np.append(1., abar[:-1])
        with np.errstate(divide="ignore", invalid="ignore"):
            post_var = (1 - self.v_posterior) * betas * (1. - prev) / (1. - abar) + self.v_posterior * betas
            coef1 = betas * np.sqrt(prev) / (1. - abar)
            coef2 = (1. - prev) * np.sqrt(1. - betas) / (1. - abar)
        zeros = torch.zeros(self.num_timesteps)
        if self.parameterization != "v":
            self.register_buffer("sqrt_recip_alphas_cumprod", f32(np.sqrt(1. / abar)))
            self.register_buffer("sqrt_recipm1_alphas_cumprod", f32(np.sqrt(1. / abar - 1)))
        else:
            self.register_buffer("sqrt_recip_alphas_cumprod", zeros.clone())
            self.register_buffer("sqrt_recipm1_alphas_cumprod", zeros.clone())
        self.register_buffer("posterior_variance", f32(post_var))
        self.register_buffer("posterior_log_variance_clipped", f32(np.log(np.maximum(post_var, 1e-20))))
        self.register_buffer("posterior_mean_coef1", f32(coef1))
        self.register_buffer("posterior_mean_coef2", f32(coef2))
        # ddpm3d.py:173-186 (training only, not persistent): weights of the vlb term — ones for v-prediction
        if self.parameterization == "eps":
            with np.errstate(divide="ignore", invalid="ignore"):
                lvlb = betas ** 2 / (2 * post_var * (1. - betas) * (1. - abar))
        elif self.parameterization == "x0":
            lvlb = 0.5 * np.sqrt(abar) / (2. * 1 - abar)
        else:
            lvlb = np.ones_like(betas)
        lvlb = np.array(lvlb, dtype=np.float64)
        lvlb[0] = lvlb[1]
        self.register_buffer("lvlb_weights", f32(lvlb), persistent=False)

    def init_from_ckpt(self, path, ignore_keys=list(), only_model=False):
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        return (self.model if only_model else self).load_state_dict(sd, strict=False)

    # ---- v-parameterisation (ddpm3d.py:239-251): elementwise with per-sample schedule scalars -> one HIP launch each
    def _combine(self, ca, x, cb, y, t):
        from mudg_amd import ops
        return ops.lincomb(x, y, ca.to(x.device)[t], cb.to(x.device)[t])

    def predict_start_from_z_and_v(self, x_t, t, v):
        return self._combine(self.sqrt_alphas_cumprod, x_t, -self.sqrt_one_minus_alphas_cumprod, v, t)

    def predict_eps_from_z_and_v(self, x_t, t, v):
        return self._combine(self.sqrt_alphas_cumprod, v, self.sqrt_one_minus_alphas_cumprod, x_t, t)

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        return self._combine(self.sqrt_alphas_cumprod, x_start, self.sqrt_one_minus_alphas_cumprod, noise, t)

    def get_v(self, x, noise, t):
        # ddpm3d.py:310-314: sqrt(abar_t) noise - sqrt(1 - abar_t) x
        return self._combine(self.sqrt_alphas_cumprod, noise, -self.sqrt_one_minus_alphas_cumprod, x, t)

    # ---- training (reference ddpm3d.py:741-802, 1267-1300): the step itself lives in mudg_amd.train
    def p_losses(self, x_start, cond, t, noise=None, **kwargs):
        from mudg_amd.train import step
        return step.p_losses(self, x_start, cond, t, noise=noise, **kwargs)

    def configure_optimizers(self):
        """AdamW over the trainable UNet (+ image-projection: the Resampler trains through mudg_amd.train.resampler) parameters at
        `self.learning_rate`, as ddpm3d.py:1267-1300; the update runs on the HIP kernel (mudg_amd.train.step.AdamW has
        torch.optim.AdamW's semantics and defaults).  What the reference can also put into the optimiser but this build does not
        train raises instead of being dropped silently."""
        from mudg_amd.train import step
        if getattr(self, "cond_stage_trainable", False):
            raise NotImplementedError("cond_stage_trainable: the text tower has no HIP backward (frozen in every MuDG config)")
        if getattr(self, "learn_logvar", False):
            raise NotImplementedError("learn_logvar is off in every MuDG config")
        if getattr(self, "use_scheduler", False):
            raise NotImplementedError("use_scheduler: build the LambdaLR of configure_schedulers around the returned optimiser")
        params = [p for p in self.model.parameters() if p.requires_grad]
        proj = getattr(self, "image_proj_model", None)
        if getattr(self, "image_proj_model_trainable", False) and proj is not None:
            params.extend(proj.parameters())
        return step.AdamW(params, lr=getattr(self, "learning_rate", 1e-4))

    def training_step(self, batch, batch_idx=0):
        """`batch` = dict(x_start=latents (B, 4, T, H, W), cond={c_crossattn, c_concat}, t=(B,) long, + apply_model kwargs):
        the tensors the reference's shared_step / get_batch_input hand to p_losses (the Waymo data pipeline that makes them is
        outside the hot path).  Returns the loss; call .backward() and the optimizer as Lightning would."""
        kw = {k: v for k, v in batch.items() if k not in ("x_start", "cond", "t", "noise")}
        loss, _ = self.p_losses(batch["x_start"], batch["cond"], batch["t"], noise=batch.get("noise"), **kw)
        return loss

    def forward(self, x, c, **kwargs):
        """The training entry (ddpm3d.py:711-715): draw one timestep per sample, apply the dynamic rescale of the latents when the
        config enables it (`use_dynamic_rescale`: scale_arr[t], base_scale 0.3 / 0.7 in the MDM configs), then p_losses.
        Returns (loss, loss_dict)."""
        t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=x.device).long()
        if getattr(self, "use_dynamic_rescale", False):
            x = x * extract_into_tensor(self.scale_arr.to(x.device), t, x.shape)
        return self.p_losses(x, c, t, **kwargs)

    def shared_step(self, batch, **kwargs):
        raise NotImplementedError("get_batch_input (VAE-encoding Waymo items, CLIP towers, random conditioning dropout) is the data "
                                  "pipeline of the reference, outside the hot path: call p_losses / training_step with latents")


class LatentDiffusion(DDPM):
    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="caption",
                 cond_stage_trainable=False, cond_stage_forward=None, conditioning_key=None, uncond_prob=0.2,
                 uncond_type="empty_seq", scale_factor=1.0, scale_by_std=False, encoder_type="2d", only_model=False,
                 noise_strength=0, use_dynamic_rescale=False, base_scale=0.7, turning_step=400, interp_mode=False,
                 fps_condition_type="fs", perframe_ae=False, logdir=None, rand_cond_frame=False,
                 en_and_decode_n_samples_a_time=None, *args, **kwargs):
        self.num_timesteps_cond = default(num_timesteps_cond, 1)
        self.scale_by_std = scale_by_std
        assert self.num_timesteps_cond <= kwargs["timesteps"]
        if self.num_timesteps_cond != 1:
            raise NotImplementedError("num_timesteps_cond > 1 is not on the MuDG path")
        ckpt_path = kwargs.pop("ckpt_path", None)
        ignore_keys = kwargs.pop("ignore_keys", [])
        super().__init__(conditioning_key=default(conditioning_key, "crossattn"), *args, **kwargs)
        self.cond_stage_trainable, self.cond_stage_key = cond_stage_trainable, cond_stage_key
        self.noise_strength, self.use_dynamic_rescale = noise_strength, use_dynamic_rescale
        self.interp_mode, self.fps_condition_type, self.perframe_ae = interp_mode, fps_condition_type, perframe_ae
        self.logdir, self.rand_cond_frame = logdir, rand_cond_frame
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time
        ddc = _cfg_get(_cfg_get(first_stage_config, "params"), "ddconfig")
        mult = _cfg_get(ddc, "ch_mult")
        self.num_downs = len(mult) - 1 if mult is not None else 0
        if scale_by_std:
            self.register_buffer("scale_factor", torch.tensor(scale_factor))
        else:
            self.scale_factor = scale_factor
        self.base_scale, self.turning_step = base_scale, turning_step
        if use_dynamic_rescale:
            self._register_scale_arr()
        self.instantiate_first_stage(first_stage_config)
        self.instantiate_cond_stage(cond_stage_config)
        self.first_stage_config, self.cond_stage_config = first_stage_config, cond_stage_config
        self.clip_denoised = False
        self.cond_stage_forward = cond_stage_forward
        assert encoder_type in ["2d", "3d"]
        self.encoder_type = encoder_type
        self.uncond_prob, self.classifier_free_guidance = uncond_prob, uncond_prob > 0
        assert uncond_type in ["zero_embed", "empty_seq"]
        self.uncond_type = uncond_type
        self.restarted_from_ckpt = False
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys, only_model=only_model)
            self.restarted_from_ckpt = True

    def _register_scale_arr(self):
        # ddpm3d.py:522-527: 1 -> base_scale over `turning_step` steps, then flat (length turning_step + T)
        arr = np.concatenate((np.linspace(1.0, self.base_scale, self.turning_step),
                              np.full(self.num_timesteps, self.base_scale)))
        self.register_buffer("scale_arr", torch.tensor(arr, dtype=torch.float32, device="cpu"))

    def rebuild_schedules(self, device=None):
        """Recompute every schedule buffer from the stored hyper-parameters (they are pure functions of the config).
        Needed after constructing under torch.device('meta') + to_empty(), where buffers carry no data."""
        self.register_schedule(given_betas=self.given_betas, beta_schedule=self.beta_schedule,
                               timesteps=self.timesteps, linear_start=self.linear_start,
                               linear_end=self.linear_end, cosine_s=self.cosine_s)
        if self.use_dynamic_rescale:
            self._register_scale_arr()
        if device is not None:
            for name, buf in list(self.named_buffers(recurse=False)):
                self.register_buffer(name, buf.to(device))
        return self

    def _freeze(self, model):
        model = model.eval()
        model.train = disabled_train.__get__(model)
        for p in model.parameters():
            p.requires_grad = False
        return model

    def instantiate_first_stage(self, config):
        self.first_stage_model = self._freeze(instantiate_from_config(config))

    def instantiate_cond_stage(self, config):
        model = instantiate_from_config(config)
        self.cond_stage_model = model if self.cond_stage_trainable else self._freeze(model)

    def get_learned_conditioning(self, c):
        """Text embedding via the configured cond stage (OpenCLIP in MuDG's configs — outside this path's scope; any
        module with .encode / __call__ returning (B, 77, D) works)."""
        m = self.cond_stage_model
        if self.cond_stage_forward is None:
            return m.encode(c) if callable(getattr(m, "encode", None)) else m(c)
        return getattr(m, self.cond_stage_forward)(c)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        """scale_factor * posterior.sample() (ddpm3d.py:611-618); the scale rides in the sampling kernel."""
        if isinstance(encoder_posterior, torch.Tensor):
            from mudg_amd import ops
            return ops.lincomb(encoder_posterior, encoder_posterior,
                               torch.full((encoder_posterior.shape[0],), float(self.scale_factor), device=encoder_posterior.device),
                               torch.zeros(encoder_posterior.shape[0], device=encoder_posterior.device))
        return encoder_posterior.sample(noise=noise, scale=float(self.scale_factor))

    @torch.no_grad()
    def encode_first_stage(self, x):
        """(B, 3, T, H, W) or (N, 3, H, W) pixels -> scaled latents (ddpm3d.py:620-644).  With perframe_ae the
        reference encodes and samples frame by frame; the CPU noise draws happen in that same order here, while the
        encoder itself runs on batches of frames (they are independent)."""
        five = x.dim() == 5
        if five:
            b, c, t, h, w = x.shape
            frames = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        else:
            frames = x
        posterior = self.first_stage_model.encode(frames)
        n, c2, hh, ww = posterior.parameters.shape
        if self.perframe_ae:
            noise = torch.cat([torch.randn((1, c2 // 2, hh, ww)) for _ in range(n)], 0)
        else:
            noise = torch.randn((n, c2 // 2, hh, ww))
        z = self.get_first_stage_encoding(posterior, noise=noise)
        if five:
            z = z.reshape(b, t, c2 // 2, hh, ww).permute(0, 2, 1, 3, 4)
        return z

    @torch.no_grad()
    def decode_core(self, z, **kwargs):
        """z (B, C, T, h, w) or (N, C, h, w) latents -> pixels; divides by scale_factor and decodes frame by frame
        (perframe_ae) — every frame is independent, so batching only changes launch counts."""
        from mudg_amd.engine import vae
        return vae.decode_latents(self.first_stage_model, z, 1.0 / float(self.scale_factor), self.perframe_ae)

    def decode_first_stage(self, z, **kwargs):
        return self.decode_core(z, **kwargs)

    differentiable_decode_first_stage = decode_first_stage

    def apply_model(self, x_noisy, t, cond, **kwargs):
        if not isinstance(cond, dict):
            cond = {("c_concat" if self.model.conditioning_key == "concat" else "c_crossattn"):
                    cond if isinstance(cond, list) else [cond]}
        label = kwargs.get("class_label", None)
        if label is None:
            raise TypeError("apply_model needs class_label=(B, 1) (0 colour / 500 depth / 1 semantic)")
        out = self.model(x_noisy, t, label[:, 0], **cond, **kwargs)
        return out[0] if isinstance(out, tuple) else out


class LatentVisualDiffusion(LatentDiffusion):
    def __init__(self, img_cond_stage_config, image_proj_stage_config, freeze_embedder=True,
                 image_proj_model_trainable=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.image_proj_model_trainable = image_proj_model_trainable
        self.embedder = instantiate_from_config(img_cond_stage_config)
        if freeze_embedder:
            self.embedder = self._freeze(self.embedder)
        self.image_proj_model = instantiate_from_config(image_proj_stage_config)
        if not image_proj_model_trainable:
            self.image_proj_model = self._freeze(self.image_proj_model)
