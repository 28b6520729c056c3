"""Shared test helpers: golden fixtures, seeded weights, error metrics."""
import importlib.util
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    spec = importlib.util.spec_from_file_location("golden_" + name, os.path.join(GOLDEN, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


seeding = _load("seeding")
cfgs = _load("configs")


def golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=True)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def seeded_sd(param_shapes, seed, expect_checksum=None):
    sd = seeding.seeded_state_dict({k: tuple(v) for k, v in param_shapes.items()}, seed)
    if expect_checksum is not None:
        got = seeding.checksum(sd)
        assert abs(got - expect_checksum) <= 1e-9 * abs(expect_checksum), (got, expect_checksum)
    return sd


def unet_inputs(cfg, shp, seed):
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    x = seeding.seeded_input("x", (B, cfg["in_channels"], T, H, W), seed)
    ctx = seeding.seeded_input("context", (B, 77 + 16 * T, cfg["context_dim"]), seed)
    return x, ctx


def pipeline_inputs(g, steps=None):
    """Rebuild the tensors make_golden.golden_pipeline fed the reference."""
    shp, s, seed = g["shape"], dict(g["sampler"]), g["seed"] + 2
    if steps is not None:
        s["steps"] = steps
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    cd = g["unet_cfg"]["context_dim"]
    return {
        "ctx_c": seeding.seeded_input("ctx_cond", (B, 77 + 16 * T, cd), seed),
        "ctx_u": seeding.seeded_input("ctx_uncond", (B, 77 + 16 * T, cd), seed),
        "concat": seeding.seeded_input("c_concat", (B, 8, T, H, W), seed, 0.18215 * 5),
        "x_T": seeding.seeded_input("x_T", (B, 4, T, H, W), seed),
        "noises": [seeding.seeded_input(f"noise{i}", (B, 4, T, H, W), seed) for i in range(s["steps"])],
        "class_label": torch.tensor(s["class_labels"], dtype=torch.long)[:, None],
        "fs": torch.full((B,), s["fs"], dtype=torch.long),
    }
