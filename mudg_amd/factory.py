"""Model construction helpers shared by bench.py, the smoke test and the driver."""
import torch

from . import configs, synthetic


def build_synthetic_model(resolution="1024", device="cuda", seed=123, overrides=None):
    """LatentVisualDiffusion (UNet + AutoencoderKL, placeholder conditioners) with synthetic weights on `device`.
    Parameters are created on the meta device and materialised directly in HBM — the 1.44 B-parameter UNet never
    exists in host memory."""
    from lvdm.models.ddpm3d import LatentVisualDiffusion
    kwargs = configs.latent_visual_diffusion(resolution)
    if overrides:
        kwargs.update(overrides)
    with torch.device("meta"):
        model = LatentVisualDiffusion(**kwargs)
    model.to_empty(device=device)
    synthetic.fill_synthetic(model, seed)
    model.rebuild_schedules(device)
    return model.eval()


def synthetic_inputs(model, resolution="1024", batch=1, device="cuda", seed=123, latent_shape=None, context_dim=1024):
    """x_T, hybrid conditioning (cond / uncond) and labels of the shapes MuDG's driver produces
    (virtual_pose_render.py:90-100): c_concat (B, 8, T, h, w), c_crossattn (B, 77 + 16 T, D)."""
    c, t, h, w = latent_shape or configs.LATENT_SHAPE[str(resolution)]
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=device)
    concat = rn(batch, 8, t, h, w) * (0.18215 * 5)
    cond = {"c_crossattn": [rn(batch, 77 + 16 * t, context_dim)], "c_concat": [concat]}
    uc = {"c_crossattn": [rn(batch, 77 + 16 * t, context_dim)], "c_concat": [concat]}
    labels = torch.tensor([0, 500, 1], device=device)[torch.arange(batch, device=device) % 3][:, None]
    return {"x_T": rn(batch, c, t, h, w), "cond": cond, "uc": uc, "class_label": labels,
            "fs": torch.full((batch,), 10, dtype=torch.long, device=device), "sparse_x": concat[:, :4]}
