#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE implementation (read-only at /root/reference).

Run in the build container only:   python tests/golden/make_golden.py
The reference is imported with /root/reference on sys.path and /root/repo NOT on it (both trees own a package
called `lvdm`); missing third-party packages (cv2, pytorch_lightning, torchvision) are replaced by harness stubs
that carry no arithmetic.  Only inputs, outputs and a weight checksum are stored — weights are re-derived from
seeding.py on both sides.  The fixtures written next to this file are data, not code.
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MUDG_REFERENCE", "/root/reference")
assert os.path.isdir(REF), "the reference tree is required to regenerate goldens"
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

torch.set_grad_enabled(False)
torch.set_num_threads(max(1, os.cpu_count() or 1))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


seeding = _load("seeding")
cfgs = _load("configs")

# ---------------------------------------------------------------------------------------------- stubs (no math)
sys.modules["cv2"] = types.ModuleType("cv2")
pl = types.ModuleType("pytorch_lightning")


class _LightningModule(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device

    def log(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass


pl.LightningModule = _LightningModule
plu = types.ModuleType("pytorch_lightning.utilities")
plu.rank_zero_only = lambda f: f
pl.utilities = plu
sys.modules["pytorch_lightning"] = pl
sys.modules["pytorch_lightning.utilities"] = plu
tv = types.ModuleType("torchvision")
tvu = types.ModuleType("torchvision.utils")
tvu.make_grid = lambda *a, **k: None
tv.utils = tvu
sys.modules["torchvision"] = tv
sys.modules["torchvision.utils"] = tvu

stubs = types.ModuleType("golden_stubs")


class DummyEmbedder(nn.Module):
    """Stands in for the CLIP text/image encoders, which are outside the path."""

    def __init__(self, **kw):
        super().__init__()
        self.p = nn.Parameter(torch.zeros(1))

    def encode(self, x):
        raise RuntimeError("conditioning is injected as tensors in the goldens")

    def forward(self, x):
        raise RuntimeError("conditioning is injected as tensors in the goldens")


stubs.DummyEmbedder = DummyEmbedder
sys.modules["golden_stubs"] = stubs


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return AttrDict(v) if isinstance(v, dict) else v


from lvdm.modules.networks.openaimodel3d import UNetModel           # noqa: E402  (the reference's)
from lvdm.models.utils_diffusion import timestep_embedding          # noqa: E402
from lvdm.models.samplers import ddim as ref_ddim                   # noqa: E402
from lvdm.models.ddpm3d import LatentVisualDiffusion                # noqa: E402


def reseed(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items() if v.dtype.is_floating_point}
    sd = seeding.seeded_state_dict(shapes, seed)
    full = module.state_dict()
    full.update(sd)
    module.load_state_dict(full, strict=True)
    return shapes, seeding.checksum(sd)


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


# ---------------------------------------------------------------------------------------------- 1. schedule KATs
def golden_schedule():
    out = {}
    for base in (0.3, 0.7):
        cfg = dict(cfgs.DIFFUSION, base_scale=base)
        model = build_diffusion(cfgs.UNET_B, cfg)
        sampler = CPUSampler(model)
        entry = {"betas": model.betas.clone(), "alphas_cumprod": model.alphas_cumprod.clone(),
                 "sqrt_alphas_cumprod": model.sqrt_alphas_cumprod.clone(),
                 "sqrt_one_minus_alphas_cumprod": model.sqrt_one_minus_alphas_cumprod.clone(),
                 "scale_arr": model.scale_arr.clone()}
        for steps in (50, 2):
            for spacing in ("uniform_trailing", "uniform"):
                sampler.make_schedule(steps, ddim_discretize=spacing, ddim_eta=1.0, verbose=False)
                entry[f"ddim_{steps}_{spacing}"] = {
                    "timesteps": torch.as_tensor(np.ascontiguousarray(sampler.ddim_timesteps)),
                    "alphas": torch.as_tensor(sampler.ddim_alphas).clone(),
                    "alphas_prev": torch.as_tensor(np.asarray(sampler.ddim_alphas_prev, dtype=np.float64)),
                    "sigmas": torch.as_tensor(sampler.ddim_sigmas).double().clone(),
                    "sqrt_one_minus_alphas": torch.as_tensor(sampler.ddim_sqrt_one_minus_alphas).clone(),
                    "scale_arr": sampler.ddim_scale_arr.clone(), "scale_arr_prev": sampler.ddim_scale_arr_prev.clone()}
        out[f"base_{base}"] = entry
    t = torch.tensor([999, 19, 10, 500, 0, 1])
    out["timestep_embedding_320"] = {"t": t, "emb": timestep_embedding(t, 320)}
    out["timestep_embedding_64"] = {"t": t, "emb": timestep_embedding(t, 64)}
    save("schedule.pt", out)


# ---------------------------------------------------------------------------------------------- 2. UNet forward
def unet_inputs(cfg, shp, seed):
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    x = seeding.seeded_input("x", (B, cfg["in_channels"], T, H, W), seed)
    ctx = seeding.seeded_input("context", (B, 77 + 16 * T, cfg["context_dim"]), seed)
    return x, ctx


def golden_unet(tag, cfg, shp):
    net = UNetModel(**cfg).eval()
    shapes, cks = reseed(net, cfgs.SEED)
    x, ctx = unet_inputs(cfg, shp, cfgs.SEED)
    B = shp["B"]
    cases = []
    for t, labels in ((999, [0, 500, 1]), (19, [1, 0, 500])):
        ts = torch.full((B,), t, dtype=torch.long)
        lab = torch.tensor(labels[:B], dtype=torch.long)
        fs = torch.full((B,), 10, dtype=torch.long)
        y = net(x, ts, c_label=lab, context=ctx, fs=fs)
        cases.append({"t": ts, "c_label": lab, "fs": fs, "y": y.clone()})
    save(f"unet_{tag}.pt", {"cfg": cfg, "shape": shp, "seed": cfgs.SEED, "checksum": cks,
                           "param_shapes": shapes, "cases": cases})


# ---------------------------------------------------------------------------------------------- 3. sampler + decode
class CPUSampler(ref_ddim.DDIMSampler):
    """The reference sampler with its hard-coded .to('cuda') removed (ddim.py:18-22); arithmetic untouched."""

    def register_buffer(self, name, attr):
        setattr(self, name, attr)


def build_diffusion(unet_cfg, diff_cfg):
    dummy = {"target": "golden_stubs.DummyEmbedder", "params": {}}
    model = LatentVisualDiffusion(
        img_cond_stage_config=dummy, image_proj_stage_config=dummy,
        first_stage_config=AttrDict({"target": "lvdm.models.autoencoder.AutoencoderKL",
                                     "params": {"embed_dim": 4, "ddconfig": cfgs.VAE_DD,
                                                "lossconfig": {"target": "torch.nn.Identity"}}}),
        cond_stage_config=dummy,
        unet_config=AttrDict({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": unet_cfg}),
        **diff_cfg)
    return model.eval()


def golden_pipeline():
    s = cfgs.SAMPLER
    model = build_diffusion(cfgs.UNET_B, cfgs.DIFFUSION)
    unet_shapes, unet_cks = reseed(model.model.diffusion_model, cfgs.SEED)
    vae_shapes, vae_cks = reseed(model.first_stage_model, cfgs.SEED + 1)
    shp = cfgs.UNET_B_SHAPE
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    seed = cfgs.SEED + 2
    ctx_c = seeding.seeded_input("ctx_cond", (B, 77 + 16 * T, cfgs.UNET_B["context_dim"]), seed)
    ctx_u = seeding.seeded_input("ctx_uncond", (B, 77 + 16 * T, cfgs.UNET_B["context_dim"]), seed)
    concat = seeding.seeded_input("c_concat", (B, 8, T, H, W), seed, 0.18215 * 5)
    x_T = seeding.seeded_input("x_T", (B, 4, T, H, W), seed)
    noises = [seeding.seeded_input(f"noise{i}", (B, 4, T, H, W), seed) for i in range(s["steps"])]
    class_label = torch.tensor(s["class_labels"], dtype=torch.long)[:, None]
    fs = torch.full((B,), s["fs"], dtype=torch.long)
    cond = {"c_crossattn": [ctx_c], "c_concat": [concat]}
    uc = {"c_crossattn": [ctx_u], "c_concat": [concat]}

    trace, it = [], iter(noises)
    ref_ddim.noise_like = lambda shape, device, repeat=False: next(it)          # inject recorded noise
    orig_apply = model.apply_model
    outs = []

    def tapped(x, t, c, **kw):
        y = orig_apply(x, t, c, **kw)
        outs.append(y.clone())
        return y

    model.apply_model = tapped
    sampler = CPUSampler(model)
    orig_p = sampler.p_sample_ddim

    def p_tapped(x, c, t, index, **kw):
        xp, x0 = orig_p(x, c, t, index=index, **kw)
        trace.append({"t": t.clone(), "index": index, "e_c": outs[-2], "e_u": outs[-1], "x_prev": xp.clone(),
                      "pred_x0": x0.clone()})
        return xp, x0

    sampler.p_sample_ddim = p_tapped
    samples, _ = sampler.sample(S=s["steps"], conditioning=cond, batch_size=B, shape=[4, T, H, W], verbose=False,
                                unconditional_guidance_scale=s["cfg_scale"], unconditional_conditioning=uc,
                                eta=s["eta"], cfg_img=None, mask=None, x0=None, fs=fs, x_T=x_T,
                                timestep_spacing=s["spacing"], guidance_rescale=s["guidance_rescale"],
                                sparse_x=concat[:, :4], class_label=class_label,
                                unconditional_conditioning_img_nonetext=None)
    decoded = model.decode_first_stage(samples)
    z1 = seeding.seeded_input("z_dec", (2, 4, 8, 8), seed)
    dec_direct = model.first_stage_model.decode(z1)
    save("pipeline.pt", {
        "unet_cfg": cfgs.UNET_B, "diffusion_cfg": cfgs.DIFFUSION, "vae_ddconfig": cfgs.VAE_DD, "sampler": s,
        "shape": shp, "seed": cfgs.SEED, "unet_checksum": unet_cks, "vae_checksum": vae_cks,
        "unet_param_shapes": unet_shapes, "vae_param_shapes": vae_shapes,
        "ddim_timesteps": torch.as_tensor(np.ascontiguousarray(sampler.ddim_timesteps)),
        "trace": trace, "samples": samples.clone(), "decoded": decoded.clone(),
        "decode_direct": {"z": z1, "out": dec_direct.clone()}})


def golden_encode():
    """AutoencoderKL.encode moments + encode_first_stage (perframe, CPU-generator noise) on the tiny VAE."""
    model = build_diffusion(cfgs.UNET_B, cfgs.DIFFUSION)
    shapes, cks = reseed(model.first_stage_model, cfgs.SEED + 1)
    x = seeding.seeded_input("pixels", (2, 3, 3, 64, 64), cfgs.SEED + 3, 0.5).clamp(-1, 1)       # (B, 3, T, H, W)
    frames = x.permute(0, 2, 1, 3, 4).reshape(6, 3, 64, 64)
    moments = model.first_stage_model.encode(frames).parameters
    torch.manual_seed(777)
    z = model.encode_first_stage(x)
    save("encode.pt", {"vae_ddconfig": cfgs.VAE_DD, "seed": cfgs.SEED, "vae_checksum": cks, "vae_param_shapes": shapes,
                       "moments": moments.clone(), "z": z.clone(), "cpu_seed": 777,
                       "scale_factor": cfgs.DIFFUSION["scale_factor"]})


def golden_resampler():
    from lvdm.modules.encoders.resampler import Resampler
    net = Resampler(**cfgs.RESAMPLER).eval()
    shapes, cks = reseed(net, cfgs.SEED + 4)
    x = seeding.seeded_input("clip_tokens", (3, 257, cfgs.RESAMPLER["embedding_dim"]), cfgs.SEED + 4)
    save("resampler.pt", {"cfg": cfgs.RESAMPLER, "seed": cfgs.SEED + 4, "checksum": cks, "param_shapes": shapes,
                          "out": net(x).clone()})


def golden_postprocess():
    """Per-modality post-processing of decoded frames (eval_tools.py:20-27 uint8 conversion, 71-72 depth = channel mean,
    309-347 visualize_semantic = nearest of 19 palette colours): integer / byte work, captured from the reference's own
    function (visualize_semantic) and its own inline expressions."""
    tvio = types.ModuleType("torchvision.io")
    sys.modules["torchvision"].io = tvio
    sys.modules["torchvision.io"] = tvio
    import PIL.Image                                                          # eval_tools.py uses PIL.Image after a bare `import PIL`
    spec = importlib.util.spec_from_file_location("ref_eval_tools", os.path.join(REF, "virtual_render", "eval_tools.py"))
    et = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(et)
    g = torch.Generator().manual_seed(cfgs.SEED + 9)
    video = torch.randn(2, 3, 3, 24, 32, generator=g) * 0.9                  # (b, c, t, h, w); values beyond [-1, 1] included
    video[0, :, 0, 0, :4] = torch.tensor([[-1.0, 1.0, 0.0, 0.999999], [-1.5, 1.5, -0.0, 0.003921568], [1.0, -1.0, 0.5, -0.5]])
    clamped = torch.clamp(video.float(), -1., 1.)                           # eval_tools.py:23
    grids = [((clamped[i] + 1.0) / 2.0 * 255).to(torch.uint8).permute(1, 2, 3, 0) for i in range(2)]   # eval_tools.py:26-27, thwc
    u8 = torch.stack(grids)                                                 # (b, t, h, w, c)
    depth = torch.stack([torch.stack([torch.mean(gr[t].permute(2, 0, 1).float(), dim=0, keepdim=True) / 255
                                      for t in range(gr.shape[0])]) for gr in grids])            # eval_tools.py:71
    pal = torch.tensor([[255, 120, 50], [255, 192, 203], [255, 255, 0], [0, 150, 245], [0, 255, 255], [255, 127, 0], [255, 0, 0],
                        [255, 240, 150], [135, 60, 0], [160, 32, 240], [255, 0, 255], [139, 137, 137], [75, 0, 75], [150, 240, 80],
                        [230, 230, 250], [0, 175, 0], [0, 255, 127], [222, 155, 161], [140, 62, 69]], dtype=torch.uint8)
    img = torch.randint(0, 256, (3, 24, 32), generator=g, dtype=torch.uint8)
    img[:, 0, :19] = pal.t()                                                # every palette colour exactly
    img[:, 1, 0] = torch.tensor([255, 123, 25], dtype=torch.uint8)          # equidistant from entries 0 and 5 -> first wins?
    vis, lab = et.visualize_semantic(img, return_pt=True)
    save("postprocess.pt", {"video": video, "u8": u8, "depth": depth, "semantic_in": img, "semantic_vis": vis.clone(),
                            "semantic_labels": lab.clone()})


# ---------------------------------------------------------------------------------------------- 8. 50-step runs
def _pipeline_inputs(steps):
    shp = cfgs.UNET_B_SHAPE
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    seed = cfgs.SEED + 2
    d = {"ctx_c": seeding.seeded_input("ctx_cond", (B, 77 + 16 * T, cfgs.UNET_B["context_dim"]), seed),
         "ctx_u": seeding.seeded_input("ctx_uncond", (B, 77 + 16 * T, cfgs.UNET_B["context_dim"]), seed),
         "concat": seeding.seeded_input("c_concat", (B, 8, T, H, W), seed, 0.18215 * 5),
         "x_T": seeding.seeded_input("x_T", (B, 4, T, H, W), seed),
         "noises": [seeding.seeded_input(f"noise{i}", (B, 4, T, H, W), seed) for i in range(steps)]}
    return d, (B, T, H, W)


def _seeded_diffusion():
    model = build_diffusion(cfgs.UNET_B, cfgs.DIFFUSION)
    unet_shapes, unet_cks = reseed(model.model.diffusion_model, cfgs.SEED)
    vae_shapes, vae_cks = reseed(model.first_stage_model, cfgs.SEED + 1)
    return model, {"unet_cfg": cfgs.UNET_B, "diffusion_cfg": cfgs.DIFFUSION, "vae_ddconfig": cfgs.VAE_DD,
                   "shape": cfgs.UNET_B_SHAPE, "seed": cfgs.SEED, "unet_checksum": unet_cks, "vae_checksum": vae_cks,
                   "unet_param_shapes": unet_shapes, "vae_param_shapes": vae_shapes}


def golden_pipeline50():
    """The 2-step pipeline golden at the reference's real step count: 50 DDIM steps, eta 1 with recorded noise, and an
    eta = 0 run; every 5th x_prev is kept so error growth along the trajectory can be seen."""
    s = cfgs.SAMPLER50
    model, meta = _seeded_diffusion()
    inp, (B, T, H, W) = _pipeline_inputs(s["steps"])
    class_label = torch.tensor(s["class_labels"], dtype=torch.long)[:, None]
    fs = torch.full((B,), s["fs"], dtype=torch.long)
    cond = {"c_crossattn": [inp["ctx_c"]], "c_concat": [inp["concat"]]}
    uc = {"c_crossattn": [inp["ctx_u"]], "c_concat": [inp["concat"]]}
    runs = {}
    for eta in (1.0, 0.0):
        it = iter(inp["noises"])
        ref_ddim.noise_like = lambda shape, device, repeat=False: next(it)
        sampler = CPUSampler(model)
        kept = []
        orig_p = sampler.p_sample_ddim

        def p_tapped(x, c, t, index, **kw):
            xp, x0 = orig_p(x, c, t, index=index, **kw)
            if index % 5 == 0:
                kept.append({"index": index, "x_prev": xp.clone()})
            return xp, x0

        sampler.p_sample_ddim = p_tapped
        samples, _ = sampler.sample(S=s["steps"], conditioning=cond, batch_size=B, shape=[4, T, H, W], verbose=False,
                                    unconditional_guidance_scale=s["cfg_scale"], unconditional_conditioning=uc,
                                    eta=eta, cfg_img=None, mask=None, x0=None, fs=fs, x_T=inp["x_T"],
                                    timestep_spacing=s["spacing"], guidance_rescale=s["guidance_rescale"],
                                    sparse_x=inp["concat"][:, :4], class_label=class_label,
                                    unconditional_conditioning_img_nonetext=None)
        runs[f"eta{eta:g}"] = {"eta": eta, "kept": kept, "samples": samples.clone(),
                               "decoded": model.decode_first_stage(samples).clone(),
                               "ddim_timesteps": torch.as_tensor(np.ascontiguousarray(sampler.ddim_timesteps))}
    save("pipeline50.pt", dict(meta, sampler=s, runs=runs))


# ---------------------------------------------------------------------------------------------- 9. three-way CFG
def golden_threeway():
    """ddim_multiplecond.DDIMSampler (p_sample_ddim 213-236): e_u + cfg_img (e_img - e_u) + s (e_c - e_img), 2 steps."""
    from lvdm.models.samplers import ddim_multiplecond as ref_mc

    class CPUSampler3(ref_mc.DDIMSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    s = cfgs.THREEWAY
    model, meta = _seeded_diffusion()
    inp, (B, T, H, W) = _pipeline_inputs(s["steps"])
    class_label = torch.tensor(s["class_labels"], dtype=torch.long)[:, None]
    fs = torch.full((B,), s["fs"], dtype=torch.long)
    n_img = 16 * T
    cond = {"c_crossattn": [inp["ctx_c"]], "c_concat": [inp["concat"]]}
    uc = {"c_crossattn": [inp["ctx_u"]], "c_concat": [inp["concat"]]}
    # image tokens of the conditional context, text tokens of the unconditional one (virtual_pose_render.py:106-109)
    uc2 = {"c_crossattn": [torch.cat([inp["ctx_u"][:, :77], inp["ctx_c"][:, 77:77 + n_img]], 1)], "c_concat": [inp["concat"]]}
    it = iter(inp["noises"])
    ref_mc.noise_like = lambda shape, device, repeat=False: next(it)
    outs, trace = [], []
    orig_apply = model.apply_model

    def tapped(x, t, c, **kw):
        y = orig_apply(x, t, c, **kw)
        outs.append(y.clone())
        return y

    model.apply_model = tapped
    sampler = CPUSampler3(model)
    orig_p = sampler.p_sample_ddim

    def p_tapped(x, c, t, index, **kw):
        xp, x0 = orig_p(x, c, t, index=index, **kw)
        trace.append({"index": index, "e_c": outs[-3], "e_u": outs[-2], "e_m": outs[-1], "x_prev": xp.clone(),
                      "pred_x0": x0.clone()})
        return xp, x0

    sampler.p_sample_ddim = p_tapped
    samples, _ = sampler.sample(S=s["steps"], conditioning=cond, batch_size=B, shape=[4, T, H, W], verbose=False,
                                unconditional_guidance_scale=s["cfg_scale"], unconditional_conditioning=uc,
                                eta=s["eta"], cfg_img=s["cfg_img"], mask=None, x0=None, fs=fs, x_T=inp["x_T"],
                                timestep_spacing=s["spacing"], guidance_rescale=s["guidance_rescale"],
                                sparse_x=inp["concat"][:, :4], class_label=class_label,
                                unconditional_conditioning_img_nonetext=uc2)
    save("threeway.pt", dict(meta, sampler=s, trace=trace, samples=samples.clone()))


# ---------------------------------------------------------------------------------------------- 10. the driver function
def golden_driver():
    """The reference's own image_guided_synthesis (virtual_pose_render.py:62-147) run on the tiny model with the fake CLIP
    towers of towers.py: Resampler, two VAE encodes (CPU-generator posterior noise), cond / uc / uc_2 assembly, two-way and
    three-way guided sampling with x_T and the per-step noise injected, decode."""
    towers = _load("towers")
    for name in ("omegaconf", "megfile", "torchvision.transforms", "virtual_render.eval_tools", "virtual_render.data_tools"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["omegaconf"].OmegaConf = object
    sys.modules["megfile"].smart_open = None
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["pytorch_lightning"].seed_everything = lambda *a, **k: None
    for n in ("save_virtual_color_results", "save_virtual_depth_results", "save_virtual_semantic_results"):
        setattr(sys.modules["virtual_render.eval_tools"], n, None)
    for n in ("get_color_frames", "get_sparse_depth", "get_depth_frames", "get_semantic_frames"):
        setattr(sys.modules["virtual_render.data_tools"], n, None)
    spec = importlib.util.spec_from_file_location("ref_vpr", os.path.join(REF, "virtual_render", "virtual_pose_render.py"))
    vpr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vpr)
    from lvdm.models.samplers import ddim_multiplecond as ref_mc
    from lvdm.modules.encoders.resampler import Resampler

    class CPUSampler3(ref_mc.DDIMSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    vpr.DDIMSampler, vpr.DDIMSampler_multicond = CPUSampler, CPUSampler3
    d = cfgs.DRIVER
    model, meta = _seeded_diffusion()
    shp = cfgs.UNET_B_SHAPE
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    model.image_proj_model = Resampler(**d["resampler"]).eval()
    rs_shapes, rs_cks = reseed(model.image_proj_model, cfgs.SEED + 5)
    model.embedder = towers.FakeImageTower(d["clip_tokens"], d["clip_dim"], d["tower_seed_img"])
    model.cond_stage_model = towers.FakeTextTower(cfgs.UNET_B["context_dim"], d["tower_seed_txt"])
    seed = cfgs.SEED + 6
    px = d["pixels"]
    sparse = seeding.seeded_input("drv_sparse", (B, 3, T, px, px), seed, 0.5).clamp(-1, 1)
    depth = seeding.seeded_input("drv_depth", (B, 3, T, px, px), seed, 0.5).clamp(-1, 1)
    x_T = seeding.seeded_input("drv_x_T", (B, 4, T, H, W), seed)
    noises = [seeding.seeded_input(f"drv_noise{i}", (B, 4, T, H, W), seed) for i in range(2)]
    labels = torch.tensor(cfgs.SAMPLER["class_labels"], dtype=torch.long)[:, None]
    common = dict(ddim_steps=2, ddim_eta=1.0, unconditional_guidance_scale=cfgs.SAMPLER["cfg_scale"], fs=cfgs.SAMPLER["fs"],
                  text_input=True, timestep_spacing=cfgs.SAMPLER["spacing"], guidance_rescale=cfgs.SAMPLER["guidance_rescale"])
    outs = {}
    for tag, extra in (("two_way", {}), ("three_way", {"multiple_cond_cfg": True, "cfg_img": d["cfg_img"]})):
        it = iter(noises)
        ref_ddim.noise_like = lambda shape, device, repeat=False: next(it)
        ref_mc.noise_like = ref_ddim.noise_like
        torch.manual_seed(d["cpu_seed"])
        out = vpr.image_guided_synthesis(model, ["a street"] * B, sparse, depth, labels, [B, 4, T, H, W], x_T=x_T,
                                         **common, **extra)
        outs[tag] = out.clone()
    save("driver.pt", dict(meta, driver=d, resampler_param_shapes=rs_shapes, resampler_checksum=rs_cks, outs=outs))


# ---------------------------------------------------------------------------------------------- 10b. sampler options
class DampingCorrector:
    """A stand-in score corrector (the reference ships none): shrinks eps and pulls it towards the latent."""

    def modify_score(self, model, e_t, x, t, c, gain=0.9, pull=0.05):
        return gain * e_t + pull * x


def golden_sampler_options():
    """The DDIMSampler options MuDG's own driver leaves at their defaults, each run on the reference with recorded noise:
    mask blending (noised and clean original), a `timesteps` prefix of the schedule, the full-schedule "original steps"
    walk, temperature, an eps-parameterised model (+ score corrector), decode and stochastic_encode."""
    s = cfgs.SAMPLER
    model, meta = _seeded_diffusion()
    inp, (B, T, H, W) = _pipeline_inputs(8)
    seed = cfgs.SEED + 7
    class_label = torch.tensor(s["class_labels"], dtype=torch.long)[:, None]
    fs = torch.full((B,), s["fs"], dtype=torch.long)
    cond = {"c_crossattn": [inp["ctx_c"]], "c_concat": [inp["concat"]]}
    uc = {"c_crossattn": [inp["ctx_u"]], "c_concat": [inp["concat"]]}
    x0_lat = seeding.seeded_input("opt_x0", (B, 4, T, H, W), seed)
    mask = (seeding.seeded_input("opt_mask", (B, 1, T, H, W), seed) > 0).float()
    q_noises = [seeding.seeded_input(f"opt_q{i}", (B, 4, T, H, W), seed) for i in range(8)]
    common = dict(unconditional_guidance_scale=s["cfg_scale"], unconditional_conditioning=uc, fs=fs,
                  guidance_rescale=s["guidance_rescale"], sparse_x=inp["concat"][:, :4], class_label=class_label,
                  unconditional_conditioning_img_nonetext=None)
    orig_q = model.q_sample

    def fresh(steps, eta=1.0, spacing=s["spacing"]):
        it = iter(inp["noises"])
        ref_ddim.noise_like = lambda shape, device, repeat=False: next(it)
        qn = iter(q_noises)
        model.q_sample = lambda x_start, t, noise=None: orig_q(x_start, t, noise=next(qn))
        sampler = CPUSampler(model)
        sampler.make_schedule(ddim_num_steps=steps, ddim_discretize=spacing, ddim_eta=eta, verbose=False)
        return sampler

    out = {}
    shape = (B, 4, T, H, W)
    for tag, extra in (("mask", {}), ("mask_clean", {"clean_cond": True})):
        x, _ = fresh(3).ddim_sampling(cond, shape, x_T=inp["x_T"], mask=mask, x0=x0_lat, verbose=False, **extra, **common)
        out[tag] = x.clone()
    x, inter = fresh(8).ddim_sampling(cond, shape, x_T=inp["x_T"], timesteps=5, verbose=False, log_every_t=1, **common)
    out["subset"] = {"x": x.clone(), "n_inter": len(inter["x_inter"])}
    x, _ = fresh(8).ddim_sampling(cond, shape, x_T=inp["x_T"], ddim_use_original_steps=True, timesteps=3, verbose=False,
                                  **common)
    out["original"] = x.clone()
    x, _ = fresh(3).ddim_sampling(cond, shape, x_T=inp["x_T"], temperature=0.6, verbose=False, **common)
    out["temperature"] = x.clone()
    # eps-prediction divides by sqrt(a_t): with zero terminal SNR the trailing schedule's first step (t = 999) is 0/0 in the
    # reference, so these two cases walk the "uniform" schedule (4 steps, t <= 751)
    model.parameterization = "eps"
    x, _ = fresh(4, spacing="uniform").ddim_sampling(cond, shape, x_T=inp["x_T"], verbose=False, **common)
    out["eps"] = x.clone()
    x, _ = fresh(4, spacing="uniform").ddim_sampling(cond, shape, x_T=inp["x_T"], verbose=False, score_corrector=DampingCorrector(),
                                  corrector_kwargs={"gain": 0.8, "pull": 0.1}, **common)
    out["eps_corrected"] = x.clone()
    model.parameterization = "v"
    # decode() passes neither fs nor the driver's extra kwargs
    # (MuDG's apply_model insists on a class_label keyword, which decode() has no way to pass: supplied by a wrapper here,
    # and the same wrapper in the test)
    orig_apply = model.apply_model
    model.apply_model = lambda x, t, c, **kw: orig_apply(x, t, c, **dict({"class_label": class_label, "fs": fs}, **kw))
    x = fresh(6).decode(inp["x_T"], cond, 4, unconditional_guidance_scale=s["cfg_scale"], unconditional_conditioning=uc)
    model.apply_model = orig_apply
    out["decode"] = x.clone()
    smp = fresh(6)
    t_idx = torch.tensor([1, 4, 5][:B], dtype=torch.long)
    out["encode"] = {"t": t_idx, "ddim": smp.stochastic_encode(x0_lat, t_idx, noise=q_noises[0]).clone(),
                     "original": smp.stochastic_encode(x0_lat, t_idx * 100, use_original_steps=True, noise=q_noises[1]).clone()}
    model.q_sample = orig_q
    save("sampler_options.pt", dict(meta, sampler=s, seed_options=seed, cases=out))


# ---------------------------------------------------------------------------------------------- 11. the YAML configs
def golden_training_forward():
    """LatentDiffusion.forward — the training entry (ddpm3d.py:711-715: draw t, dynamic rescale of the latents, p_losses) — on
    the tiny pipeline model with offset noise on: the timesteps and the noise the reference drew from the seeded CPU generator
    are recorded next to the loss and its dictionary.  Evaluation mode (no dropout draw), so the value is a pure function of
    what is stored."""
    diff = dict(cfgs.DIFFUSION, noise_strength=0.1)
    model = build_diffusion(cfgs.UNET_B, diff)
    unet_shapes, unet_cks = reseed(model.model.diffusion_model, cfgs.SEED)
    shp = cfgs.UNET_B_SHAPE
    B, T, H, W = shp["B"], shp["T"], shp["H"], shp["W"]
    seed = cfgs.SEED + 9
    x = seeding.seeded_input("x_start", (B, 4, T, H, W), seed)
    ctx = seeding.seeded_input("ctx_train", (B, 77 + 16 * T, cfgs.UNET_B["context_dim"]), seed)
    concat = seeding.seeded_input("c_concat_train", (B, 8, T, H, W), seed, 0.18215 * 5)
    fs = torch.full((B,), 10, dtype=torch.long)
    class_label = torch.tensor([0, 500, 1], dtype=torch.long)[:B, None]
    cond = {"c_crossattn": [ctx], "c_concat": [concat]}
    seen = {}
    orig_q = model.q_sample

    def q_tapped(x_start, t, noise=None):
        seen.update(x_rescaled=x_start.clone(), t=t.clone(), noise=noise.clone())
        return orig_q(x_start=x_start, t=t, noise=noise)

    model.q_sample = q_tapped
    torch.manual_seed(4242)
    loss, loss_dict = model(x, cond, fs=fs, class_label=class_label)
    save("training_forward.pt", {"unet_cfg": cfgs.UNET_B, "diffusion_cfg": diff, "shape": shp, "seed": cfgs.SEED, "input_seed": seed,
                                 "cpu_seed": 4242, "unet_checksum": unet_cks, "unet_param_shapes": unet_shapes,
                                 "fs": fs, "class_label": class_label, "t": seen["t"], "noise": seen["noise"],
                                 "x_rescaled": seen["x_rescaled"], "loss": loss.clone(),
                                 "loss_dict": {k: v.clone() for k, v in loss_dict.items()}})


def golden_yaml():
    """model.params of the reference's two inference YAMLs as JSON: mudg_amd/configs.py is asserted equal to it, and every
    `target:` in it must resolve against this repo's overlay (tests/test_host_logic.py)."""
    import json
    import yaml
    out = {}
    for tag, name in (("1024", "stage2-1024_mdm_waymo_infer.yaml"), ("512", "stage1-512_mdm_waymo_infer.yaml")):
        with open(os.path.join(REF, "configs", name)) as f:
            out[tag] = yaml.safe_load(f)["model"]
    # the training configs: the model section and the trainer settings that shape one optimisation step
    for tag, name in (("train1024", "stage2-1024_mdm_waymo/config.yaml"), ("train512", "stage1-512_mdm_waymo/config.yaml")):
        with open(os.path.join(REF, "configs", name)) as f:
            cfg = yaml.safe_load(f)
        out[tag] = dict(cfg["model"], lightning={"precision": cfg["lightning"].get("precision"),
                                                 "trainer": {k: cfg["lightning"]["trainer"].get(k) for k in
                                                             ("accumulate_grad_batches", "gradient_clip_algorithm", "gradient_clip_val")}})
    path = os.path.join(HERE, "mdm_yaml.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {path}")


if __name__ == "__main__":
    if "--only-post" in sys.argv:
        golden_postprocess()
        sys.exit(0)
    if "--only-encode" in sys.argv:
        golden_encode()
        sys.exit(0)
    if "--only-resampler" in sys.argv:
        golden_resampler()
        sys.exit(0)
    if "--only-options" in sys.argv:
        golden_sampler_options()
        sys.exit(0)
    if "--only-training" in sys.argv:
        golden_training_forward()
        sys.exit(0)
    if "--only-round2" in sys.argv:
        golden_pipeline50()
        golden_threeway()
        golden_driver()
        golden_yaml()
        sys.exit(0)
    golden_schedule()
    golden_unet("a", cfgs.UNET_A, cfgs.UNET_A_SHAPE)
    golden_unet("b", cfgs.UNET_B, cfgs.UNET_B_SHAPE)
    golden_pipeline()
    golden_encode()
    golden_resampler()
    golden_postprocess()
    golden_pipeline50()
    golden_threeway()
    golden_driver()
    golden_sampler_options()
    golden_training_forward()
    golden_yaml()
