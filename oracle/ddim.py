"""oracle/ddim.py — TEST INFRASTRUCTURE.  CPU fp32 restatement of the DDIM sampler step and loop.

Follows lvdm/models/samplers/ddim.py:205-279 (p_sample_ddim), :134-203 (ddim_sampling), utils_diffusion.py:147-157
(rescale_noise_cfg) and ddpm3d.py:239-251 (v-parameterisation).  Noise is always injected by the caller so that runs
are comparable across devices (the reference draws it from the device generator).
"""
import numpy as np
import torch


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale):
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


def step_coefficients(sched, dd, index):
    """The per-step scalars, each rounded to fp32 exactly where torch.full(...) rounds them (ddim.py:251-254,262-266)."""
    t = int(dd["timesteps"][index])
    f32 = lambda v: torch.full((1,), float(v), dtype=torch.float32)
    a_prev = f32(dd["alphas_prev"][index])
    sigma = f32(dd["sigmas"][index])
    out = {
        "t": t,
        "sqrt_ac": sched["sqrt_alphas_cumprod"][t].reshape(1),
        "sqrt_1mac": sched["sqrt_one_minus_alphas_cumprod"][t].reshape(1),
        "sqrt_a_prev": a_prev.sqrt(),
        "dir_coef": (1. - a_prev - sigma ** 2).sqrt(),
        "sigma": sigma,
        "rescale": torch.ones(1),
    }
    if "scale_arr" in dd:
        out["rescale"] = f32(dd["scale_arr_prev"][index]) / f32(dd["scale_arr"][index])
    return out


def p_sample_ddim(x, e_cond, e_uncond, noise, coef, cfg_scale, guidance_rescale, e_img=None, cfg_img=None):
    """One update given the two UNet outputs (v-prediction).  Returns (x_prev, pred_x0).
    With e_img (the pass conditioned on image tokens + empty prompt) the guidance is the three-way form of
    lvdm/models/samplers/ddim_multiplecond.py:226-233: e_u + cfg_img (e_img - e_u) + s (e_c - e_img)."""
    if e_uncond is None or cfg_scale == 1.0:
        v = e_cond
    else:
        if e_img is not None:
            v = e_uncond + (cfg_scale if cfg_img is None else cfg_img) * (e_img - e_uncond) + cfg_scale * (e_cond - e_img)
        else:
            v = e_uncond + cfg_scale * (e_cond - e_uncond)
        if guidance_rescale > 0.0:
            v = rescale_noise_cfg(v, e_cond, guidance_rescale)
    e_t = coef["sqrt_ac"] * v + coef["sqrt_1mac"] * x            # predict_eps_from_z_and_v
    pred_x0 = coef["sqrt_ac"] * x - coef["sqrt_1mac"] * v        # predict_start_from_z_and_v
    pred_x0 = pred_x0 * coef["rescale"]
    dir_xt = coef["dir_coef"] * e_t
    nz = coef["sigma"] * noise if noise is not None else 0.0
    return coef["sqrt_a_prev"] * pred_x0 + dir_xt + nz, pred_x0


@torch.no_grad()
def ddim_sample(apply_model, sched, x_T, cond, uncond, steps, noises, eta=1.0, cfg_scale=7.5, guidance_rescale=0.7,
                spacing="uniform_trailing", trace=None, uncond_img=None, cfg_img=None):
    """ddim_sampling loop.  apply_model(x, t_long, cond) -> v.  noises[i] is the noise of loop iteration i.
    uncond_img: the third conditioning of ddim_multiplecond (pass order there: cond, uncond, uncond_img)."""
    from .schedule import ddim_schedule
    dd = ddim_schedule(sched, steps, spacing, eta)
    x = x_T
    b = x.shape[0]
    for i, step in enumerate(np.flip(dd["timesteps"])):
        index = steps - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        e_c = apply_model(x, ts, cond)
        e_u = apply_model(x, ts, uncond) if uncond is not None and cfg_scale != 1.0 else None
        e_m = apply_model(x, ts, uncond_img) if (e_u is not None and uncond_img is not None) else None
        coef = step_coefficients(sched, dd, index)
        x, x0 = p_sample_ddim(x, e_c, e_u, noises[i] if noises is not None else None, coef, cfg_scale, guidance_rescale,
                              e_img=e_m, cfg_img=cfg_img)
        if trace is not None:
            trace.append({"index": index, "e_c": e_c, "e_u": e_u, "e_m": e_m, "x_prev": x, "pred_x0": x0})
    return x
