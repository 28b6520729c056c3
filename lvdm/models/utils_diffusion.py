"""Schedule math of the sampler — host side, float64 numpy exactly where the reference uses it.

Reference: lvdm/models/utils_diffusion.py — make_beta_schedule 31-53, make_ddim_timesteps 56-76,
make_ddim_sampling_parameters 79-91, rescale_zero_terminal_snr 112-144, timestep_embedding 8-28.
These run once per sampling call; the per-step arithmetic they feed is the fused HIP update (mudg_ddim_step).
"""
import numpy as np
import torch


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """[N, dim] sinusoidal embedding, computed by the HIP kernel with a host-built frequency table."""
    if repeat_only:
        raise NotImplementedError("repeat_only embeddings are not used on the MuDG path")
    from mudg_amd import ops
    if not timesteps.is_cuda:
        raise RuntimeError("timestep_embedding: the MI355X path has no CPU fallback")
    return ops.timestep_embedding(timesteps, dim, max_period)


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    elif schedule == "cosine":
        steps = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        abar = torch.cos(steps / (1 + cosine_s) * np.pi / 2).pow(2)
        abar = abar / abar[0]
        betas = torch.clamp(1 - abar[1:] / abar[:-1], 0, 0.999)
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


def rescale_zero_terminal_snr(betas):
    """Shift/scale sqrt(alpha_bar) so the last timestep has exactly zero SNR (arXiv 2305.08891, Alg. 1)."""
    root = np.sqrt(np.cumprod(1.0 - betas, axis=0))
    r0, rT = root[0].copy(), root[-1].copy()
    root = (root - rT) * (r0 / (r0 - rT))
    abar = root ** 2
    return 1.0 - np.concatenate([abar[0:1], abar[1:] / abar[:-1]])


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method == "uniform":
        stride = num_ddpm_timesteps // num_ddim_timesteps
        out = np.asarray(list(range(0, num_ddpm_timesteps, stride))) + 1
    elif ddim_discr_method == "uniform_trailing":
        stride = num_ddpm_timesteps / num_ddim_timesteps
        out = np.flip(np.round(np.arange(num_ddpm_timesteps, 0, -stride))).astype(np.int64) - 1
    elif ddim_discr_method == "quad":
        out = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int) + 1
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    if verbose:
        print(f"Selected timesteps for ddim sampler: {out}")
    return out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """alphacums: fp32 tensor of cumulative alphas.  Returns (sigmas float64 tensor, alphas fp32 tensor,
    alphas_prev float64 ndarray) — the mixed precisions the reference's arithmetic produces."""
    alphacums = torch.as_tensor(alphacums).detach().cpu()
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([float(alphacums[0])] + alphacums[ddim_timesteps[:-1]].tolist())
    prev_t = torch.from_numpy(alphas_prev)
    # The reference evaluates ndarray / Tensor, which PyTorch dispatches as Tensor.reciprocal() * ndarray: the
    # reciprocal of (1 - alphas) is taken in fp32 before the float64 product.  Kept, so sigmas match bit for bit.
    ratio = (1 - alphas).reciprocal() * (1 - prev_t)
    # ... and the square root is numpy's (np.sqrt on a Tensor round-trips through ndarray), which differs from
    # torch.sqrt's vectorised CPU kernel in the last ulp for some entries.
    sigmas = eta * torch.from_numpy(np.sqrt((ratio * (1 - alphas / prev_t)).numpy()))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule "
              f"for ddim sampler {sigmas}")
    return sigmas, alphas, alphas_prev


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """Guidance rescale (reference: utils_diffusion.py:147-157): bring the guided prediction back to the per-sample
    standard deviation of the conditional one, blended by `guidance_rescale`.  The sampler's fused update does this
    inside mudg_ddim_step; this tensor form serves the score-corrector path, which needs the guided output itself."""
    axes = tuple(range(1, noise_pred_text.ndim))
    gain = noise_pred_text.std(dim=axes, keepdim=True) / noise_cfg.std(dim=axes, keepdim=True)
    return guidance_rescale * (noise_cfg * gain) + (1 - guidance_rescale) * noise_cfg
