"""Pin the CPU oracle to vectors captured from the reference itself (tests/golden/make_golden.py).
fp32 on both sides: tolerance is rel-L2 <= 1e-5 per forward and <= 1e-4 after two sampler steps + decode."""
import numpy as np
import pytest
import torch

from helpers import golden, pipeline_inputs, rel_l2, seeded_sd, unet_inputs

from oracle import ddim as o_ddim
from oracle import schedule as o_sched
from oracle import unet as o_unet
from oracle import vae as o_vae


def test_schedule_known_answers_survey_appendix_c():
    """The literal values SURVEY.md Appendix C records from the reference."""
    s = o_sched.model_schedule(base_scale=0.3)
    assert torch.allclose(s["betas"][[0, 1, 500, 998, 999]].double(),
                          torch.tensor([8.5e-04, 9.1733359337e-04, 5.5305677845e-03, 7.5113700418e-01, 1.0]).double(),
                          rtol=1e-6)
    assert torch.allclose(s["alphas_cumprod"][[0, 19, 499, 979, 998]].double(),
                          torch.tensor([0.99915, 0.98101045693, 0.24235916656, 8.5787843401e-05, 1.9678880566e-07]).double(),
                          rtol=1e-6)
    assert float(s["alphas_cumprod"][999]) == 0.0
    ts = o_sched.ddim_timesteps("uniform_trailing", 50, 1000)
    assert list(ts[:3]) == [19, 39, 59] and list(ts[-3:]) == [959, 979, 999]
    assert list(o_sched.ddim_timesteps("uniform_trailing", 2, 1000)) == [499, 999]
    dd = o_sched.ddim_schedule(s, 50, "uniform_trailing", 1.0)
    sig = torch.as_tensor(dd["sigmas"]).double()
    assert torch.allclose(sig[[0, 1, 25, 48, 49]], torch.tensor([0.0285072528, 0.1005311182, 0.3224573905,
                                                                 0.8781245652, 0.9999571052]).double(), rtol=1e-7)
    a_prev, sigma = torch.full((1,), float(dd["alphas_prev"][49])), torch.full((1,), float(dd["sigmas"][49]))
    assert float(1. - a_prev - sigma ** 2) == pytest.approx(5.9604645e-08, rel=1e-6)   # one ulp above zero
    assert float(s["scale_arr"][19]) == pytest.approx(0.9666666667, rel=1e-6) and s["scale_arr"].shape[0] == 1400


@pytest.mark.parametrize("base", [0.3, 0.7])
def test_schedule_matches_reference_buffers(base):
    g = golden("schedule.pt")[f"base_{base}"]
    s = o_sched.model_schedule(base_scale=base)
    for k in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "scale_arr"):
        assert torch.equal(s[k], g[k]), k
    for steps in (50, 2):
        for spacing in ("uniform_trailing", "uniform"):
            ref = g[f"ddim_{steps}_{spacing}"]
            dd = o_sched.ddim_schedule(s, steps, spacing, 1.0)
            assert np.array_equal(np.asarray(dd["timesteps"]), ref["timesteps"].numpy())
            assert torch.equal(torch.as_tensor(dd["alphas"]), ref["alphas"])
            assert torch.equal(torch.as_tensor(np.asarray(dd["alphas_prev"], dtype=np.float64)), ref["alphas_prev"])
            assert torch.equal(torch.as_tensor(dd["sigmas"]).double(), ref["sigmas"])
            assert torch.equal(dd["scale_arr"], ref["scale_arr"]) and torch.equal(dd["scale_arr_prev"], ref["scale_arr_prev"])


def test_sinusoid_matches_reference():
    g = golden("schedule.pt")
    for dim in (320, 64):
        e = g[f"timestep_embedding_{dim}"]
        assert torch.equal(o_unet.sinusoid(e["t"], dim), e["emb"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_unet_forward_matches_reference(tag):
    g = golden(f"unet_{tag}.pt")
    sd = seeded_sd(g["param_shapes"], g["seed"], g["checksum"])
    x, ctx = unet_inputs(g["cfg"], g["shape"], g["seed"])
    for case in g["cases"]:
        y = o_unet.unet_forward(sd, g["cfg"], x, case["t"], case["c_label"], ctx, case["fs"])
        assert y.shape == case["y"].shape
        assert rel_l2(y, case["y"]) < 1e-5
        assert float(case["y"].abs().max()) > 1e-3        # a dead (all-zero) golden would pin nothing


def test_sampler_and_decode_match_reference():
    g = golden("pipeline.pt")
    usd = seeded_sd(g["unet_param_shapes"], g["seed"], g["unet_checksum"])
    vsd = seeded_sd(g["vae_param_shapes"], g["seed"] + 1, g["vae_checksum"])
    inp, s, dc = pipeline_inputs(g), g["sampler"], g["diffusion_cfg"]
    sched = o_sched.model_schedule(dc["timesteps"], dc["linear_start"], dc["linear_end"], dc["rescale_betas_zero_snr"],
                                   dc["base_scale"])
    lab = inp["class_label"][:, 0]

    def apply_model(x, t, ctx):
        xc = torch.cat([x, inp["concat"]], dim=1)                      # DiffusionWrapper 'hybrid', ddpm3d.py:1317-1319
        return o_unet.unet_forward(usd, g["unet_cfg"], xc, t, lab, ctx, inp["fs"])

    trace = []
    samples = o_ddim.ddim_sample(apply_model, sched, inp["x_T"], inp["ctx_c"], inp["ctx_u"], s["steps"], inp["noises"],
                                 s["eta"], s["cfg_scale"], s["guidance_rescale"], s["spacing"], trace)
    for got, ref in zip(trace, g["trace"]):
        assert rel_l2(got["e_c"], ref["e_c"]) < 1e-4 and rel_l2(got["e_u"], ref["e_u"]) < 1e-4
        assert rel_l2(got["pred_x0"], ref["pred_x0"]) < 1e-4 and rel_l2(got["x_prev"], ref["x_prev"]) < 1e-4
    assert rel_l2(samples, g["samples"]) < 1e-4
    # the update alone, fed the reference's own UNet outputs: isolates p_sample_ddim arithmetic
    dd = o_sched.ddim_schedule(sched, s["steps"], s["spacing"], s["eta"])
    x = inp["x_T"]
    for i, ref in enumerate(g["trace"]):
        coef = o_ddim.step_coefficients(sched, dd, ref["index"])
        xp, x0 = o_ddim.p_sample_ddim(x, ref["e_c"], ref["e_u"], inp["noises"][i], coef, s["cfg_scale"], s["guidance_rescale"])
        assert rel_l2(x0, ref["pred_x0"]) < 2e-6 and rel_l2(xp, ref["x_prev"]) < 2e-6
        x = ref["x_prev"]
    dec = o_vae.decode_first_stage(vsd, g["vae_ddconfig"], g["samples"], dc["scale_factor"])
    assert rel_l2(dec, g["decoded"]) < 1e-5
    d2 = o_vae.decode(vsd, g["vae_ddconfig"], g["decode_direct"]["z"])
    assert rel_l2(d2, g["decode_direct"]["out"]) < 1e-5


def _oracle_pipeline(g, steps):
    usd = seeded_sd(g["unet_param_shapes"], g["seed"], g["unet_checksum"])
    vsd = seeded_sd(g["vae_param_shapes"], g["seed"] + 1, g["vae_checksum"])
    inp, dc = pipeline_inputs(g, steps), g["diffusion_cfg"]
    sched = o_sched.model_schedule(dc["timesteps"], dc["linear_start"], dc["linear_end"], dc["rescale_betas_zero_snr"],
                                   dc["base_scale"])
    lab = inp["class_label"][:, 0]

    def apply_model(x, t, ctx):
        return o_unet.unet_forward(usd, g["unet_cfg"], torch.cat([x, inp["concat"]], dim=1), t, lab, ctx, inp["fs"])

    return usd, vsd, inp, sched, apply_model


@pytest.mark.parametrize("eta", [1.0, 0.0])
def test_fifty_step_sampler_and_decode_match_reference(eta):
    """The reference's real step count (render.sh:25-31): 50 guided steps with the recorded noise (eta 1) and eta 0.
    fp32 on both sides with different summation orders: the trajectory error is printed at every kept step."""
    g = golden("pipeline50.pt")
    s, run = g["sampler"], g["runs"][f"eta{eta:g}"]
    usd, vsd, inp, sched, apply_model = _oracle_pipeline(g, s["steps"])
    trace = []
    samples = o_ddim.ddim_sample(apply_model, sched, inp["x_T"], inp["ctx_c"], inp["ctx_u"], s["steps"], inp["noises"],
                                 eta, s["cfg_scale"], s["guidance_rescale"], s["spacing"], trace)
    by_index = {t["index"]: t for t in trace}
    errs = [(k["index"], rel_l2(by_index[k["index"]]["x_prev"], k["x_prev"])) for k in run["kept"]]
    dec = o_vae.decode_first_stage(vsd, g["vae_ddconfig"], samples, g["diffusion_cfg"]["scale_factor"])
    e_s, e_d = rel_l2(samples, run["samples"]), rel_l2(dec, run["decoded"])
    print(f"oracle 50 steps eta {eta:g}: samples {e_s:.2e} decoded {e_d:.2e}; along the way " +
          " ".join(f"{i}:{e:.1e}" for i, e in errs))
    assert e_s < 1e-4 and e_d < 1e-4


def test_three_way_guidance_matches_reference():
    """ddim_multiplecond.DDIMSampler.p_sample_ddim (213-236) via the oracle: per-pass outputs, the update and the samples."""
    g = golden("threeway.pt")
    s = g["sampler"]
    usd, vsd, inp, sched, apply_model = _oracle_pipeline(g, s["steps"])
    uc2 = torch.cat([inp["ctx_u"][:, :77], inp["ctx_c"][:, 77:]], 1)
    trace = []
    samples = o_ddim.ddim_sample(apply_model, sched, inp["x_T"], inp["ctx_c"], inp["ctx_u"], s["steps"], inp["noises"],
                                 s["eta"], s["cfg_scale"], s["guidance_rescale"], s["spacing"], trace, uncond_img=uc2,
                                 cfg_img=s["cfg_img"])
    for got, ref in zip(trace, g["trace"]):
        for k in ("e_c", "e_u", "e_m", "pred_x0", "x_prev"):
            assert rel_l2(got[k], ref[k]) < 1e-4, k
    assert rel_l2(samples, g["samples"]) < 1e-4
    dd = o_sched.ddim_schedule(sched, s["steps"], s["spacing"], s["eta"])
    x = inp["x_T"]
    for i, ref in enumerate(g["trace"]):       # the update alone on the reference's own three outputs
        coef = o_ddim.step_coefficients(sched, dd, ref["index"])
        xp, x0 = o_ddim.p_sample_ddim(x, ref["e_c"], ref["e_u"], inp["noises"][i], coef, s["cfg_scale"],
                                      s["guidance_rescale"], e_img=ref["e_m"], cfg_img=s["cfg_img"])
        assert rel_l2(x0, ref["pred_x0"]) < 2e-6 and rel_l2(xp, ref["x_prev"]) < 2e-6
        x = ref["x_prev"]


def test_vae_encode_matches_reference():
    from helpers import seeding
    g = golden("encode.pt")
    vsd = seeded_sd(g["vae_param_shapes"], g["seed"] + 1, g["vae_checksum"])
    x = seeding.seeded_input("pixels", (2, 3, 3, 64, 64), g["seed"] + 3, 0.5).clamp(-1, 1)
    frames = x.permute(0, 2, 1, 3, 4).reshape(6, 3, 64, 64)
    mom = o_vae.encode(vsd, g["vae_ddconfig"], frames)
    assert rel_l2(mom, g["moments"]) < 1e-5
    torch.manual_seed(g["cpu_seed"])                       # perframe_ae: one CPU randn per frame, in frame order
    noise = torch.cat([torch.randn(1, 4, 8, 8) for _ in range(6)], 0)
    z = o_vae.posterior_sample(g["moments"], noise, g["scale_factor"]).reshape(2, 3, 4, 8, 8).permute(0, 2, 1, 3, 4)
    assert rel_l2(z, g["z"]) < 1e-6


def test_resampler_matches_reference():
    from helpers import seeding
    from oracle import resampler as o_res
    g = golden("resampler.pt")
    sd = seeded_sd(g["param_shapes"], g["seed"], g["checksum"])
    x = seeding.seeded_input("clip_tokens", (3, 257, g["cfg"]["embedding_dim"]), g["seed"])
    out = o_res.forward(sd, x, g["cfg"]["heads"], g["cfg"]["depth"])
    assert out.shape == g["out"].shape and rel_l2(out, g["out"]) < 1e-5


def test_postprocess_oracle_matches_reference_bit_for_bit():
    """uint8 conversion, depth = channel mean and the 19-colour semantic lookup (eval_tools.py) are byte / integer work:
    the numpy restatement must reproduce the reference's outputs exactly."""
    import numpy as np
    from oracle import postprocess as pp
    g = golden("postprocess.pt")
    u8 = pp.frames_to_uint8(g["video"].numpy())
    assert u8.dtype == np.uint8 and np.array_equal(u8, g["u8"].numpy())
    depth = pp.depth_from_uint8(g["u8"].numpy())
    assert np.array_equal(depth, g["depth"].numpy())
    vis, lab = pp.visualize_semantic(g["semantic_in"].numpy())
    assert np.array_equal(lab, g["semantic_labels"].numpy()) and np.array_equal(vis, g["semantic_vis"].numpy())
    assert lab[0, :19].tolist() == list(range(19))          # every palette colour maps to itself
