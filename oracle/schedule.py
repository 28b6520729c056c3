"""oracle/schedule.py — TEST INFRASTRUCTURE.  CPU restatement of the diffusion schedule math.

Follows lvdm/models/utils_diffusion.py:31-53 (make_beta_schedule), :56-76 (make_ddim_timesteps), :79-91
(make_ddim_sampling_parameters), :112-144 (rescale_zero_terminal_snr), lvdm/models/ddpm3d.py:123-186
(register_schedule) and :522-527 (dynamic-rescale scale_arr).  float64 numpy throughout, cast to float32 where the
reference casts.
"""
import numpy as np
import torch


def linear_betas(n, linear_start, linear_end):
    # utils_diffusion.py:32-35: linspace(sqrt(a), sqrt(b), n, float64) ** 2
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64) ** 2).numpy()


def zero_terminal_snr(betas):
    # utils_diffusion.py:112-144 (Algorithm 1 of arXiv 2305.08891)
    abar_sqrt = np.sqrt(np.cumprod(1.0 - betas, axis=0))
    first, last = abar_sqrt[0].copy(), abar_sqrt[-1].copy()
    abar_sqrt = (abar_sqrt - last) * (first / (first - last))
    abar = abar_sqrt ** 2
    alphas = np.concatenate([abar[0:1], abar[1:] / abar[:-1]])
    return 1.0 - alphas


def model_schedule(timesteps=1000, linear_start=0.00085, linear_end=0.012, zero_snr=True, base_scale=0.3,
                   turning_step=400, dynamic_rescale=True):
    """The buffers DDPM.register_schedule / LatentDiffusion.__init__ create (fp32 tensors)."""
    betas = linear_betas(timesteps, linear_start, linear_end)
    if zero_snr:
        betas = zero_terminal_snr(betas)
    ac = np.cumprod(1.0 - betas, axis=0)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    out = {
        "betas": f32(betas),
        "alphas_cumprod": f32(ac),
        "alphas_cumprod_prev": f32(np.append(1.0, ac[:-1])),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
    }
    if dynamic_rescale:
        out["scale_arr"] = f32(np.concatenate((np.linspace(1.0, base_scale, turning_step),
                                               np.full(timesteps, base_scale))))
    return out


def ddim_timesteps(method, n_ddim, n_ddpm):
    # utils_diffusion.py:56-76
    if method == "uniform":
        return np.asarray(list(range(0, n_ddpm, n_ddpm // n_ddim))) + 1
    if method == "uniform_trailing":
        return np.flip(np.round(np.arange(n_ddpm, 0, -(n_ddpm / n_ddim)))).astype(np.int64) - 1
    if method == "quad":
        return ((np.linspace(0, np.sqrt(n_ddpm * .8), n_ddim)) ** 2).astype(int) + 1
    raise NotImplementedError(method)


def ddim_schedule(sched, n_ddim, method="uniform_trailing", eta=1.0):
    """What DDIMSampler.make_schedule derives (lvdm/models/samplers/ddim.py:24-57): note the mixed dtypes the
    reference carries — alphas fp32 tensor, alphas_prev float64 numpy, sigmas float64 tensor."""
    ts = ddim_timesteps(method, n_ddim, sched["alphas_cumprod"].shape[0])
    ac = sched["alphas_cumprod"].cpu()
    alphas = ac[ts]                                                        # fp32 tensor
    alphas_prev = np.asarray([ac[0]] + ac[ts[:-1]].tolist())               # float64 numpy of fp32 values
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))   # float64 tensor
    out = {"timesteps": ts, "alphas": alphas, "alphas_prev": alphas_prev, "sigmas": sigmas,
           "sqrt_one_minus_alphas": np.sqrt(1.0 - alphas)}
    if "scale_arr" in sched:
        sa = sched["scale_arr"].cpu()[ts]
        out["scale_arr"] = sa
        out["scale_arr_prev"] = torch.cat([sa[0:1], sa[:-1]])
    return out
