"""Parity at BASELINE.json's full size (MDM1024: latents (B, 4, 16, 72, 128), 1.44 B-parameter UNet) through properties
that need no reference output: the CPU oracle takes minutes per forward at this size, so the HIP path is checked against
itself where the mathematics says two computations must agree.

* clip independence — the claim behind clip-level data parallelism: the UNet output of clip a must not
  depend on what else is in the batch;
* determinism — no atomics, fixed reduction orders: repeated runs are bit-identical, eager and hipGraph replay alike;
* the fused DDIM update is affine in the injected noise and reproduces x when every coefficient is the identity;
* VAE decode is per frame."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(cuda):
    from mudg_amd import factory
    model = factory.build_synthetic_model("1024", cuda, seed=7)
    inp = factory.synthetic_inputs(model, "1024", 2, cuda, seed=11)
    return model, inp


def _forward(model, inp, sl, t=601):
    x = inp["x_T"][sl]
    cond = {"c_crossattn": [inp["cond"]["c_crossattn"][0][sl]], "c_concat": [inp["cond"]["c_concat"][0][sl]]}
    ts = torch.full((x.shape[0],), t, device=x.device, dtype=torch.long)
    with torch.no_grad():
        return model.apply_model(x, ts, cond, fs=inp["fs"][sl], class_label=inp["class_label"][sl])


def test_unet_clips_are_independent_and_runs_are_deterministic_at_mdm1024(big):
    model, inp = big
    both = _forward(model, inp, slice(0, 2))
    assert both.shape == (2, 4, 16, 72, 128) and torch.isfinite(both).all()
    again = _forward(model, inp, slice(0, 2))
    assert torch.equal(both, again)                                   # bit-reproducible
    a = _forward(model, inp, slice(0, 1))
    b = _forward(model, inp, slice(1, 2))
    for got, want in ((both[0:1], a), (both[1:2], b)):
        rel = ((got - want).float().norm() / want.float().norm()).item()
        assert rel < 1e-6, rel                                        # same arithmetic whatever shares the launch
    other_t = _forward(model, inp, slice(0, 1), t=41)
    assert ((other_t - a).float().norm() / a.float().norm()).item() > 1e-3       # and it does depend on the timestep


def test_hipgraph_replay_is_bit_identical_to_eager_at_mdm1024(big):
    model, inp = big
    unet = model.model.diffusion_model
    eager = _forward(model, inp, slice(0, 1))
    unet.use_hip_graph = True
    try:
        first = _forward(model, inp, slice(0, 1))
        second = _forward(model, inp, slice(0, 1))
    finally:
        unet.use_hip_graph = False
    assert torch.equal(first, eager) and torch.equal(second, eager)


def test_fused_ddim_update_is_affine_in_the_noise_at_mdm1024(big):
    from lvdm.models.samplers.ddim import DDIMSampler
    from mudg_amd import ops
    model, inp = big
    sampler = DDIMSampler(model)
    sampler.make_schedule(50, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    coef = sampler.step_coefficients(30, 7.5, 0.7)
    g = torch.Generator(device=inp["x_T"].device).manual_seed(3)
    x = inp["x_T"][:1].contiguous()
    e_c, e_u, n1, n2 = (torch.randn(x.shape, generator=g, device=x.device) for _ in range(4))
    s1, p1 = ops.ddim_step(x, e_c, e_u, n1, coef)
    s2, p2 = ops.ddim_step(x, e_c, e_u, n2, coef)
    sm, pm = ops.ddim_step(x, e_c, e_u, 0.5 * (n1 + n2), coef)
    assert torch.equal(p1, p2) and torch.equal(p1, pm)               # pred_x0 does not see the noise
    assert ((0.5 * (s1 + s2) - sm).norm() / sm.norm()).item() < 1e-6


def test_vae_decode_is_per_frame_at_576x1024(big):
    model, inp = big
    z = inp["x_T"][:1, :, :2].contiguous() * 0.5                      # two latent frames (1, 4, 2, 72, 128)
    with torch.no_grad():
        pair = model.decode_first_stage(z)
        f0 = model.decode_first_stage(z[:, :, :1].contiguous())
        f1 = model.decode_first_stage(z[:, :, 1:].contiguous())
    assert pair.shape == (1, 3, 2, 576, 1024) and torch.isfinite(pair).all()
    for got, want in ((pair[:, :, :1], f0), (pair[:, :, 1:], f1)):
        assert ((got - want).float().norm() / want.float().norm()).item() < 1e-6


def test_three_modality_batch_at_mdm512_matches_single_clips(big):
    """The reference's own batch shape (colour / depth / semantic, virtual_pose_render.py:90-100) at MDM512 (40 x 64
    latents: other tile counts, other kernel variants than MDM1024): every stream equals its single-clip run."""
    from mudg_amd import factory
    model, _ = big
    inp = factory.synthetic_inputs(model, "512", 3, model.betas.device if hasattr(model, "betas") else "cuda", seed=5)
    both = _forward(model, inp, slice(0, 3), t=333)
    assert both.shape == (3, 4, 16, 40, 64) and torch.isfinite(both).all()
    for i in range(3):
        one = _forward(model, inp, slice(i, i + 1), t=333)
        rel = ((both[i:i + 1] - one).float().norm() / one.float().norm()).item()
        assert rel < 1e-6, (i, rel)


def test_guided_ddim_steps_are_clip_independent_at_mdm1024(big):
    """Two guided DDIM steps (CFG 7.5, rescale 0.7, eta 1 with injected noise) on a batch of two clips against the same
    steps on each clip alone: the per-sample statistics of the fused update (std over C, T, H, W) and everything
    upstream of them must not mix clips."""
    from lvdm.models.samplers.ddim import DDIMSampler
    model, inp = big
    sampler = DDIMSampler(model)
    sampler.make_schedule(50, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    dev = inp["x_T"].device
    g = torch.Generator(device=dev).manual_seed(17)
    noises = [torch.randn(inp["x_T"].shape, generator=g, device=dev) for _ in range(2)]

    def run(sl):
        x = inp["x_T"][sl].clone()
        pick = lambda d: {"c_crossattn": [d["c_crossattn"][0][sl]], "c_concat": [d["c_concat"][0][sl]]}
        cond, uc = pick(inp["cond"]), pick(inp["uc"])
        for step, index in enumerate((49, 48)):
            ts = torch.full((x.shape[0],), int(sampler.ddim_timesteps[index]), device=dev, dtype=torch.long)
            import lvdm.models.samplers.ddim as ddim_mod
            orig = ddim_mod.noise_like
            ddim_mod.noise_like = lambda shape, device, repeat=False, n=noises[step][sl]: n
            try:
                with torch.no_grad():
                    x = sampler.p_sample_ddim(x, cond, ts, index=index, unconditional_guidance_scale=7.5,
                                              unconditional_conditioning=uc, guidance_rescale=0.7, fs=inp["fs"][sl],
                                              class_label=inp["class_label"][sl])[0]
            finally:
                ddim_mod.noise_like = orig
        return x

    both = run(slice(0, 2))
    assert torch.isfinite(both).all()
    for i in range(2):
        one = run(slice(i, i + 1))
        rel = ((both[i:i + 1] - one).float().norm() / one.float().norm()).item()
        assert rel < 1e-6, (i, rel)


def _mode():
    """Operand mode of this process, with the fp8-score switch of BASELINE config 5 as its own name."""
    import os
    from mudg_amd import hip
    return hip.operand_name() + ("+fp8scores" if os.environ.get("MUDG_ATTN_FP8") == "1" else "")


def _forward_vs_oracle(model, res, seed, frames=None):
    import time
    from helpers import cached_oracle, record_parity
    from mudg_amd import configs, factory, hip
    from oracle import unet as o_unet
    unet = model.model.diffusion_model
    dev = next(unet.parameters()).device
    shape = None
    if frames is not None:                        # a shorter clip at the same spatial size (77 + 16 x frames context tokens)
        c, _, h, w = configs.LATENT_SHAPE[res]
        shape = (c, frames, h, w)
    inp = factory.synthetic_inputs(model, res, 1, dev, seed=seed, latent_shape=shape)
    x = torch.cat([inp["x_T"], inp["cond"]["c_concat"][0]], dim=1)
    ts = torch.full((1,), 499, device=dev, dtype=torch.long)
    lab, fs, ctx = inp["class_label"][:, 0], inp["fs"], inp["cond"]["c_crossattn"][0]
    with torch.no_grad():
        got = unet(x, ts, c_label=lab, context=ctx, fs=fs).float().cpu()

    def oracle():
        sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
        cfg = dict(configs.UNET_MDM) if frames is None else dict(configs.UNET_MDM, temporal_length=frames)
        return o_unet.unet_forward(sd, cfg, x.cpu(), ts.cpu(), lab.cpu(), ctx.cpu(), fs.cpu(), head_chunk=8)

    tag = "" if frames is None else f"_{frames}frames"
    flop = configs.UNET_TFLOP[res] * (1.0 if frames is None else frames / 16.0)
    t0 = time.perf_counter()
    want, hit = cached_oracle(f"unet_forward_mdm{res}{tag}_model7_seed{seed}_t499", oracle)
    dt = time.perf_counter() - t0
    err = ((got - want).double().norm() / want.double().norm()).item()
    tol = {"bf16": 2.5e-2, "fp16": 4e-3, "bf16x3": 2e-4, "bf16x6": 2e-5}[hip.operand_name()]
    took = "cached from an earlier process of this test run" if hit else \
        f"{dt:.1f} s on {torch.get_num_threads()} threads = {flop / dt:.3f} TFLOP/s"
    what = "full-size" if frames is None else f"{frames}-frame"
    print(f"[{_mode()}] MDM{res} {what} UNet forward vs CPU oracle: rel-L2 {err:.3e} (bound {tol:g}); oracle {took}")
    record_parity(_mode(), f"mdm{res}{tag}_unet_forward_vs_cpu_oracle", err)
    assert got.shape == want.shape and err < tol


def _step_reference(weights_model, res, frames, seed, decode_frames):
    """The CPU-oracle side of ONE guided DDIM step (S = 1, uniform_trailing -> t = 999; CFG 7.5, rescale 0.7, eta 0: conditional +
    unconditional UNet forward, the fused update) on MDM<res> latents (1, 4, frames, h, w) and the decode of the first `decode_frames`
    frames, memoised on disk (helpers.cached_oracle) for every test and operand-mode child that compares against it:
        {"samples", "decoded", "e_c"}      e_c = the oracle's CONDITIONAL UNet output of that step — the reference of the forward-only
                                           tests, which used to pay for an oracle forward of their own (a minute each).
    Weights come from `weights_model` (synthetic, seed 7: the UNet and VAE weights do not depend on the resolution the model object was
    built for), the schedule constants from the `res` configuration.  -> (inputs on the GPU, reference dict, cache hit, seconds)."""
    import time
    from helpers import cached_oracle
    from mudg_amd import configs, factory
    from oracle import ddim as o_ddim, schedule as o_sched, unet as o_unet, vae as o_vae
    unet = weights_model.model.diffusion_model
    dev = next(unet.parameters()).device
    c, _, h, w = configs.LATENT_SHAPE[res]
    inp = factory.synthetic_inputs(weights_model, res, 1, dev, seed=seed, latent_shape=(c, frames, h, w))

    def oracle():
        usd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
        vsd = {k: v.detach().float().cpu() for k, v in weights_model.first_stage_model.state_dict().items()}
        kw = configs.latent_visual_diffusion(res)
        sched = o_sched.model_schedule(kw["timesteps"], kw["linear_start"], kw["linear_end"], kw["rescale_betas_zero_snr"], kw["base_scale"])
        concat, lab, fs = inp["cond"]["c_concat"][0].cpu(), inp["class_label"][:, 0].cpu(), inp["fs"].cpu()
        cfg = dict(configs.UNET_MDM, temporal_length=frames)
        apply_model = lambda x, t, ctx: o_unet.unet_forward(usd, cfg, torch.cat([x, concat], 1), t, lab, ctx, fs, head_chunk=8)
        trace = []
        want = o_ddim.ddim_sample(apply_model, sched, inp["x_T"].cpu(), inp["cond"]["c_crossattn"][0].cpu(), inp["uc"]["c_crossattn"][0].cpu(),
                                  1, None, 0.0, 7.5, 0.7, "uniform_trailing", trace=trace)
        want_dec = o_vae.decode_first_stage(vsd, configs.VAE_DDCONFIG, want[:, :, :decode_frames].contiguous(), kw["scale_factor"])
        return {"samples": want, "decoded": want_dec, "e_c": trace[0]["e_c"]}

    t0 = time.perf_counter()
    want, hit = cached_oracle(f"guided_step_v2_mdm{res}_{frames}frames_model7_seed{seed}_s1_eta0_{decode_frames}decoded", oracle)
    return inp, want, hit, time.perf_counter() - t0


# (resolution, frames, input seed, decoded frames) of the two default-on guided-step references
CUT512 = ("512", 16, 31, 4)
STEP1024 = ("1024", 4, 37, 1)


def _forward_vs_step_reference(model, ref):
    """One UNet forward on the HIP path — the conditional pass of the guided step `ref`: x_T with the concat channels at t = 999 —
    against the oracle's output of that same pass (_step_reference: e_c)."""
    from helpers import record_parity
    from mudg_amd import hip
    res, frames = ref[0], ref[1]
    inp, want, hit, dt = _step_reference(model, *ref)
    unet = model.model.diffusion_model
    x = torch.cat([inp["x_T"], inp["cond"]["c_concat"][0]], dim=1)
    ts = torch.full((1,), 999, device=x.device, dtype=torch.long)
    with torch.no_grad():
        got = unet(x, ts, c_label=inp["class_label"][:, 0], context=inp["cond"]["c_crossattn"][0], fs=inp["fs"]).float().cpu()
    err = ((got - want["e_c"]).double().norm() / want["e_c"].double().norm()).item()
    tol = {"bf16": 2.5e-2, "fp16": 4e-3, "bf16x3": 2e-4, "bf16x6": 2e-5}[hip.operand_name()]
    took = "cached" if hit else f"{dt:.0f} s on {torch.get_num_threads()} threads (the whole guided step + decode)"
    tag = "" if frames == 16 else f"_{frames}frames"
    what = "full-size" if frames == 16 else f"{frames}-frame"
    print(f"[{_mode()}] MDM{res} {what} UNet forward (t = 999) vs CPU oracle: rel-L2 {err:.3e} (bound {tol:g}); oracle {took}")
    record_parity(_mode(), f"mdm{res}{tag}_unet_forward_vs_cpu_oracle", err)
    assert got.shape == want["e_c"].shape and err < tol


def test_mdm512_unet_forward_matches_the_cpu_oracle_at_full_size(big):
    """Full-size NUMERICAL parity (not a property): one UNet forward of the real 1.44 B-parameter topology at MDM512
    (latents (1, 12, 16, 40, 64), context (1, 333, 1024); 12.6 TFLOP) on the HIP path against the CPU oracle on the GPU box's host
    cores (fp32 eager) — the conditional pass of the config-0 cut below, whose memoised oracle run provides the reference (round 6: the
    forward tests no longer pay for an oracle forward of their own).  Bound: the operand mode's per-forward floor.
    MUDG_SKIP_FULLSIZE_ORACLE=1 skips it."""
    import os
    if os.environ.get("MUDG_SKIP_FULLSIZE_ORACLE") == "1":
        pytest.skip("MUDG_SKIP_FULLSIZE_ORACLE=1")
    _forward_vs_step_reference(big[0], CUT512)


def test_mdm1024_4_frame_unet_forward_matches_the_cpu_oracle(big):
    """NUMERICAL parity at the benchmarked spatial size: the real 1.44 B-parameter UNet on MDM1024 latents (1, 12, T = 4, 72, 128) —
    9216-token spatial self-attention, every level-0 shape of the benchmark — with context (1, 77 + 64, 1024): 13 TFLOP; the conditional
    pass of the 4-frame guided step below (bf16x3 child: 2e-4).  MUDG_SKIP_FULLSIZE_ORACLE=1 skips it.  The 16-frame forward stays
    opt-in (four minutes)."""
    import os
    if os.environ.get("MUDG_SKIP_FULLSIZE_ORACLE") == "1":
        pytest.skip("MUDG_SKIP_FULLSIZE_ORACLE=1")
    _forward_vs_step_reference(big[0], STEP1024)


def test_mdm1024_unet_forward_matches_the_cpu_oracle_at_the_benchmark_size(big):
    """The same at the benchmarked configuration itself — MDM1024, latents (1, 12, 16, 72, 128), 52.3 TFLOP: the oracle's
    einsum attention is chunked over (frame, head) pairs to keep its 9216 x 9216 score tensors at 2.7 GB; about four
    minutes of host time.  OPT-IN (MUDG_RUN_MDM1024_ORACLE=1); the log of a run is kept under profiles/."""
    import os
    if os.environ.get("MUDG_RUN_MDM1024_ORACLE") != "1":
        pytest.skip("opt-in: MUDG_RUN_MDM1024_ORACLE=1 (four minutes of CPU oracle)")
    _forward_vs_oracle(big[0], "1024", 23)


def test_config0_two_ddim_steps_and_decode_at_mdm512_match_the_cpu_oracle(cuda, monkeypatch):
    """BASELINE.json configs[0] at its stated size: MDM512 latents (1, 4, 16, 40, 64), the real 1.44 B-parameter UNet, 2 DDIM
    steps (uniform_trailing -> t = 999, 499), CFG 7.5, rescale 0.7, eta 1 with injected noise, then the 16-frame 320 x 512
    decode — on the HIP path and on the fp32 CPU oracle (4 UNet forwards + 16 decoder frames on the host: about six
    minutes on 128 threads).  OPT-IN (MUDG_RUN_CONFIG0=1); the log of a run is kept under profiles/.  In the precision
    modes the literal 1e-3 on decoded frames is asserted — full-size numerical parity of the whole sampler + decode."""
    import os
    import time
    if os.environ.get("MUDG_RUN_CONFIG0") != "1":
        pytest.skip("opt-in: MUDG_RUN_CONFIG0=1 (six minutes of CPU oracle)")
    from lvdm.models.samplers import ddim as my_ddim
    from mudg_amd import configs, factory, hip
    from oracle import ddim as o_ddim, schedule as o_sched, unet as o_unet, vae as o_vae
    model = factory.build_synthetic_model("512", cuda, seed=7)
    inp = factory.synthetic_inputs(model, "512", 1, cuda, seed=31)
    steps = int(os.environ.get("MUDG_CONFIG0_STEPS", "2"))          # 2 = the config as stated; more = error growth at full size
    g = torch.Generator(device=cuda).manual_seed(5)
    noises = [torch.randn(inp["x_T"].shape, generator=g, device=cuda) for _ in range(steps)]
    it = iter(noises)
    monkeypatch.setattr(my_ddim, "noise_like", lambda shape, device, repeat=False: next(it))
    sampler = my_ddim.DDIMSampler(model)
    samples, _ = sampler.sample(S=steps, conditioning=inp["cond"], batch_size=1, shape=list(inp["x_T"].shape[1:]), verbose=False,
                                unconditional_guidance_scale=7.5, unconditional_conditioning=inp["uc"], eta=1.0, mask=None, x0=None,
                                fs=inp["fs"], x_T=inp["x_T"], timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                sparse_x=inp["sparse_x"], class_label=inp["class_label"], cfg_img=None,
                                unconditional_conditioning_img_nonetext=None)
    assert steps != 2 or list(sampler.ddim_timesteps) == [499, 999]
    decoded = model.decode_first_stage(samples)
    # ---- the same run on the CPU oracle
    t0 = time.perf_counter()
    unet = model.model.diffusion_model
    usd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    vsd = {k: v.detach().float().cpu() for k, v in model.first_stage_model.state_dict().items()}
    kw = configs.latent_visual_diffusion("512")
    sched = o_sched.model_schedule(kw["timesteps"], kw["linear_start"], kw["linear_end"], kw["rescale_betas_zero_snr"], kw["base_scale"])
    concat, lab, fs = inp["cond"]["c_concat"][0].cpu(), inp["class_label"][:, 0].cpu(), inp["fs"].cpu()
    apply_model = lambda x, t, ctx: o_unet.unet_forward(usd, dict(configs.UNET_MDM), torch.cat([x, concat], 1), t, lab, ctx, fs, head_chunk=8)
    want = o_ddim.ddim_sample(apply_model, sched, inp["x_T"].cpu(), inp["cond"]["c_crossattn"][0].cpu(), inp["uc"]["c_crossattn"][0].cpu(),
                              steps, [n.cpu() for n in noises], 1.0, 7.5, 0.7, "uniform_trailing")
    want_dec = o_vae.decode_first_stage(vsd, configs.VAE_DDCONFIG, want, kw["scale_factor"])
    dt = time.perf_counter() - t0
    rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
    e_s, e_d = rel(samples, want), rel(decoded, want_dec)
    print(f"[{hip.operand_name()}] config 0 at full size (MDM512, {steps} DDIM steps + 16-frame decode) vs CPU oracle: latents {e_s:.3e}  "
          f"decoded frames {e_d:.3e}; oracle {dt:.0f} s on {torch.get_num_threads()} threads")
    assert decoded.shape == (1, 3, 16, 320, 512) and torch.isfinite(decoded).all()
    from helpers import record_parity
    record_parity(hip.operand_name(), f"config0_{steps}_steps_latents_vs_cpu_oracle", e_s)
    record_parity(hip.operand_name(), f"config0_{steps}_steps_decoded_vs_cpu_oracle", e_d)
    if hip.operand_name() in ("bf16x3", "bf16x6"):
        assert e_d <= 1e-3 and e_s <= 1e-3


def _guided_step_vs_oracle(model, ref, weights_model=None):
    """One guided DDIM step + decode on the HIP path (`model`: built for ref's resolution) against _step_reference(ref).  In the
    precision modes the literal 1e-3 is asserted on latents and on the decoded frames."""
    from helpers import record_parity
    from lvdm.models.samplers import ddim as my_ddim
    from mudg_amd import hip
    res, frames, _, nd = ref
    inp, want, hit, dt = _step_reference(weights_model or model, *ref)
    sampler = my_ddim.DDIMSampler(model)
    with torch.no_grad():
        samples, _ = sampler.sample(S=1, conditioning=inp["cond"], batch_size=1, shape=list(inp["x_T"].shape[1:]), verbose=False,
                                    unconditional_guidance_scale=7.5, unconditional_conditioning=inp["uc"], eta=0.0, mask=None, x0=None,
                                    fs=inp["fs"], x_T=inp["x_T"], timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                    sparse_x=inp["sparse_x"], class_label=inp["class_label"], cfg_img=None,
                                    unconditional_conditioning_img_nonetext=None)
        assert list(sampler.ddim_timesteps) == [999]
        decoded = model.decode_first_stage(samples[:, :, :nd].contiguous())
    rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
    e_s, e_d = rel(samples, want["samples"]), rel(decoded, want["decoded"])
    took = "cached" if hit else f"{dt:.0f} s on {torch.get_num_threads()} threads"
    return e_s, e_d, took, decoded


def _check_step(e_s, e_d):
    from mudg_amd import hip
    if hip.operand_name() in ("bf16x3", "bf16x6"):
        assert e_d <= 1e-3 and e_s <= 1e-3          # THE CONTRACT (north_star: decoded frames within 1e-3 rel-L2 of the reference)
    else:
        guard = 3e-2 if hip.operand_name() == "fp16" else 1.5e-1       # regression guards, not the contract
        assert e_d < guard and e_s < guard


def _mdm1024_step(model, frames):
    from helpers import record_parity
    ref = STEP1024 if frames == 4 else ("1024", 16, 37, 1)
    e_s, e_d, took, decoded = _guided_step_vs_oracle(model, ref)
    tag = "" if frames == 16 else f"_{frames}frames"
    size = "the benchmarked size" if frames == 16 else f"the benchmarked spatial size, {frames} frames"
    print(f"[{_mode()}] MDM1024 ({size}): 1 guided DDIM step + 1-frame 576 x 1024 decode vs CPU oracle: latents {e_s:.3e}  "
          f"decoded frame {e_d:.3e}; oracle {took}")
    record_parity(_mode(), f"mdm1024{tag}_guided_step_latents_vs_cpu_oracle", e_s)
    record_parity(_mode(), f"mdm1024{tag}_guided_step_decoded_vs_cpu_oracle", e_d)
    assert decoded.shape == (1, 3, 1, 576, 1024) and torch.isfinite(decoded).all()
    _check_step(e_s, e_d)


def test_mdm1024_4_frame_guided_ddim_step_and_one_frame_decode_match_the_cpu_oracle(big):
    """THE CONTRACT AT THE BENCHMARKED SPATIAL SIZE, in the default run: the real 1.44 B-parameter UNet on MDM1024 latents
    (1, 4, T = 4, 72, 128) — every level-0 shape of the benchmark, 9216-token spatial self-attention — through ONE guided DDIM step
    (conditional + unconditional forward, CFG 7.5, rescale 0.7, the fused update) and the 576 x 1024 decode of its first frame, against
    the fp32 CPU oracle (2 x 13 TFLOP of UNet + one 5.8-TFLOP decoder frame on the host: two to three minutes, memoised for the
    operand-mode children).  The bf16x3 child (tests/test_precision_modes_gpu.py) asserts decoded frame and latents <= 1e-3 — the literal
    north_star tolerance; the 16-bit modes are held to regression guards and their errors are printed.  The 16-frame form below is
    opt-in (nine minutes of oracle).  MUDG_SKIP_FULLSIZE_ORACLE=1 skips it."""
    import os
    if os.environ.get("MUDG_SKIP_FULLSIZE_ORACLE") == "1":
        pytest.skip("MUDG_SKIP_FULLSIZE_ORACLE=1")
    _mdm1024_step(big[0], 4)


def test_mdm1024_guided_ddim_step_and_one_frame_decode_match_the_cpu_oracle(big):
    """The benchmarked configuration itself, 16 frames (two 52-TFLOP forwards + one decoder frame on the host: about nine minutes on
    128 threads).  OPT-IN (MUDG_RUN_MDM1024_STEP=1); the log of a run is kept under profiles/."""
    import os
    if os.environ.get("MUDG_RUN_MDM1024_STEP") != "1":
        pytest.skip("opt-in: MUDG_RUN_MDM1024_STEP=1 (nine minutes of CPU oracle)")
    _mdm1024_step(big[0], 16)


def test_config0_cut_one_ddim_step_and_4_frame_decode_at_mdm512_match_the_cpu_oracle(cuda):
    """A cut of BASELINE.json configs[0] that is cheap enough to run by default: MDM512 latents (1, 4, 16, 40, 64), the real
    1.44 B-parameter UNet, ONE guided DDIM step (S = 1, uniform_trailing -> t = 999; CFG 7.5, rescale 0.7, eta 0: two full
    UNet forwards + the fused update), then the decode of the first 4 frames at 320 x 512 — HIP path against the fp32 CPU
    oracle (2 UNet forwards + 4 decoder frames on the host, about two minutes on 128 threads; memoised for the forward test above and
    for the operand-mode child runs).  In the precision modes the literal 1e-3 is asserted on latents and decoded frames; the
    16-bit modes (bf16, fp16, bf16 with fp8 scores) are held to regression guards and their errors are printed."""
    import os
    if os.environ.get("MUDG_SKIP_CONFIG0_CUT") == "1":
        pytest.skip("MUDG_SKIP_CONFIG0_CUT=1")
    from helpers import record_parity
    from mudg_amd import factory
    model = factory.build_synthetic_model("512", cuda, seed=7)
    e_s, e_d, took, decoded = _guided_step_vs_oracle(model, CUT512)
    print(f"[{_mode()}] config-0 cut at full size (MDM512, 1 guided DDIM step + 4-frame decode) vs CPU oracle: latents {e_s:.3e}  "
          f"decoded frames {e_d:.3e}; oracle {took}")
    record_parity(_mode(), "config0_cut_latents_vs_cpu_oracle", e_s)
    record_parity(_mode(), "config0_cut_decoded_vs_cpu_oracle", e_d)
    assert decoded.shape == (1, 3, 4, 320, 512) and torch.isfinite(decoded).all()
    _check_step(e_s, e_d)
