"""The DDIMSampler options MuDG's own driver leaves at their defaults, on the GPU against runs of the reference with the
same recorded noise (tests/golden/sampler_options.pt, made by make_golden.golden_sampler_options): mask blending against
the noised and the clean original latent, a `timesteps` prefix of the schedule, the full-schedule "original steps" walk,
temperature, an eps-parameterised model with and without a score corrector, decode() and stochastic_encode(); plus noise
dropout (device RNG, so checked by its defining property) and x0 quantisation through a stand-in codebook.

Tolerances: latents after 3-4 guided steps, per operand mode as in test_pipeline_gpu.py (the contract constant in the
precision modes, regression guards in the 16-bit modes); the eps-parameterised cases divide by sqrt(a_t) at every step,
which amplifies UNet error ~3x — their bound is 3x wider; decode() runs four steps of CFG 7.5 without guidance rescale
(it has no such argument) — 2x.  stochastic_encode is elementwise fp32: 1e-6."""
import pytest
import torch

from helpers import golden, pipeline_inputs, rel_l2, seeding
from test_pipeline_gpu import TOL_E2E, build_model

from mudg_amd import hip as _hip

pytestmark = pytest.mark.gpu
MODE = _hip.operand_name()


class DampingCorrector:
    """The stand-in score corrector make_golden.py drove the reference with."""

    def modify_score(self, model, e_t, x, t, c, gain=0.9, pull=0.05):
        return gain * e_t + pull * x


class Setup:
    def __init__(self, cuda, monkeypatch):
        from lvdm.models.samplers import ddim as my_ddim
        self.mod, self.cuda, self.monkeypatch = my_ddim, cuda, monkeypatch
        self.g = golden("sampler_options.pt")
        self.model = build_model(self.g, cuda)
        self.inp, self.s = pipeline_inputs(self.g, steps=8), self.g["sampler"]
        shp = self.g["shape"]
        self.shape = (shp["B"], 4, shp["T"], shp["H"], shp["W"])
        seed = self.g["seed_options"]
        self.x0 = seeding.seeded_input("opt_x0", self.shape, seed).to(cuda)
        self.mask = (seeding.seeded_input("opt_mask", (self.shape[0], 1) + self.shape[2:], seed) > 0).float().to(cuda)
        self.q_noises = [seeding.seeded_input(f"opt_q{i}", self.shape, seed) for i in range(8)]
        to = lambda v: v.to(cuda)
        self.cond = {"c_crossattn": [to(self.inp["ctx_c"])], "c_concat": [to(self.inp["concat"])]}
        self.uc = {"c_crossattn": [to(self.inp["ctx_u"])], "c_concat": [to(self.inp["concat"])]}
        self.common = dict(unconditional_guidance_scale=self.s["cfg_scale"], unconditional_conditioning=self.uc,
                           fs=to(self.inp["fs"]), guidance_rescale=self.s["guidance_rescale"],
                           sparse_x=to(self.inp["concat"][:, :4]), class_label=to(self.inp["class_label"]),
                           unconditional_conditioning_img_nonetext=None)
        self.x_T = to(self.inp["x_T"])
        self.orig_q = self.model.q_sample

    def fresh(self, steps, eta=1.0, spacing=None):
        """A sampler with its schedule made and the recorded step / q_sample noise queued, as the golden script did."""
        it, qn = iter(self.inp["noises"]), iter(self.q_noises)
        self.monkeypatch.setattr(self.mod, "noise_like", lambda shape, device, repeat=False: next(it).to(device))
        self.monkeypatch.setattr(self.model, "q_sample",
                                 lambda x_start, t, noise=None: self.orig_q(x_start, t, noise=next(qn).to(x_start.device)))
        sampler = self.mod.DDIMSampler(self.model)
        sampler.make_schedule(ddim_num_steps=steps, ddim_discretize=spacing or self.s["spacing"], ddim_eta=eta, verbose=False)
        return sampler

    def check(self, name, got, want, tol):
        err = rel_l2(got, want)
        print(f"[{MODE}] sampler option {name}: rel-L2 vs reference {err:.3e}")
        assert got.shape == want.shape and err < tol, name


@pytest.fixture
def setup(cuda, monkeypatch):
    return Setup(cuda, monkeypatch)


@pytest.mark.parametrize("tag", ["mask", "mask_clean"])
def test_mask_blending_against_the_original_latent(setup, tag):
    extra = {"clean_cond": True} if tag == "mask_clean" else {}
    x, _ = setup.fresh(3).ddim_sampling(setup.cond, setup.shape, x_T=setup.x_T, mask=setup.mask, x0=setup.x0, verbose=False,
                                        **extra, **setup.common)
    setup.check(tag, x, setup.g["cases"][tag], TOL_E2E)


def test_timestep_prefix_and_original_steps_walks(setup):
    want = setup.g["cases"]["subset"]
    x, inter = setup.fresh(8).ddim_sampling(setup.cond, setup.shape, x_T=setup.x_T, timesteps=5, verbose=False,
                                            log_every_t=1, **setup.common)
    assert len(inter["x_inter"]) == want["n_inter"] == len(inter["pred_x0"])
    setup.check("timesteps prefix", x, want["x"], TOL_E2E)
    x, _ = setup.fresh(8).ddim_sampling(setup.cond, setup.shape, x_T=setup.x_T, ddim_use_original_steps=True, timesteps=3,
                                        verbose=False, **setup.common)
    setup.check("original steps", x, setup.g["cases"]["original"], TOL_E2E)


def test_temperature_scales_the_injected_noise(setup):
    x, _ = setup.fresh(3).ddim_sampling(setup.cond, setup.shape, x_T=setup.x_T, temperature=0.6, verbose=False, **setup.common)
    setup.check("temperature", x, setup.g["cases"]["temperature"], TOL_E2E)


def test_eps_parameterised_model_and_score_corrector(setup):
    setup.monkeypatch.setattr(setup.model, "parameterization", "eps")
    x, _ = setup.fresh(4, spacing="uniform").ddim_sampling(setup.cond, setup.shape, x_T=setup.x_T, verbose=False, **setup.common)
    setup.check("eps", x, setup.g["cases"]["eps"], 3 * TOL_E2E)
    x, _ = setup.fresh(4, spacing="uniform").ddim_sampling(setup.cond, setup.shape, x_T=setup.x_T, verbose=False,
                                                           score_corrector=DampingCorrector(),
                                                           corrector_kwargs={"gain": 0.8, "pull": 0.1}, **setup.common)
    setup.check("eps + corrector", x, setup.g["cases"]["eps_corrected"], 3 * TOL_E2E)
    setup.monkeypatch.setattr(setup.model, "parameterization", "v")
    with pytest.raises(AssertionError):                       # as the reference: correctors are eps-only
        setup.fresh(3).ddim_sampling(setup.cond, setup.shape, x_T=setup.x_T, verbose=False,
                                     score_corrector=DampingCorrector(), **setup.common)


def test_decode_and_stochastic_encode(setup):
    cuda = setup.cuda
    label, fs = setup.common["class_label"], setup.common["fs"]
    orig_apply = setup.model.apply_model

    def with_labels(x, t, c, **kw):      # decode() cannot pass them (see make_golden.py); the guided passes run as one doubled batch
        n = x.shape[0] // label.shape[0]
        return orig_apply(x, t, c, **dict({"class_label": label.repeat(n, 1), "fs": fs.repeat(n)}, **kw))

    setup.monkeypatch.setattr(setup.model, "apply_model", with_labels)
    x = setup.fresh(6).decode(setup.x_T, setup.cond, 4, unconditional_guidance_scale=setup.s["cfg_scale"],
                              unconditional_conditioning=setup.uc)
    setup.check("decode", x, setup.g["cases"]["decode"], 2 * TOL_E2E)     # four steps of CFG 7.5 WITHOUT guidance rescale
    want = setup.g["cases"]["encode"]
    smp = setup.fresh(6)
    t_idx = want["t"].to(cuda)
    setup.check("stochastic_encode", smp.stochastic_encode(setup.x0, t_idx, noise=setup.q_noises[0].to(cuda)), want["ddim"], 1e-6)
    setup.check("stochastic_encode (original steps)",
                smp.stochastic_encode(setup.x0, t_idx * 100, use_original_steps=True, noise=setup.q_noises[1].to(cuda)),
                want["original"], 1e-6)


def test_noise_dropout_and_quantised_x0(setup):
    """Noise dropout draws its mask from the device generator, so there is no reference vector to replay; its definition
    is checked instead: a step with dropout p equals the plain step fed noise * keep / (1 - p).  x0 quantisation needs a
    first stage with a codebook (AutoencoderKL has none — AttributeError there, as in the reference); with a stand-in
    `quantize`, x_{t-1} must follow the snapped x0."""
    cuda, mod = setup.cuda, setup.mod
    sampler = setup.fresh(3)
    ts = torch.full((setup.shape[0],), int(sampler.ddim_timesteps[1]), device=cuda, dtype=torch.long)
    noise = setup.inp["noises"][0].to(cuda)
    p = 0.25
    setup.monkeypatch.setattr(mod, "noise_like", lambda shape, device, repeat=False: noise.clone())
    torch.manual_seed(5)
    xa, _ = sampler.p_sample_ddim(setup.x_T, setup.cond, ts, index=1, noise_dropout=p, **setup.common)
    torch.manual_seed(5)
    dropped = torch.nn.functional.dropout(noise.clone(), p=p)
    assert 0.15 < float((dropped == 0).float().mean()) < 0.35
    setup.monkeypatch.setattr(mod, "noise_like", lambda shape, device, repeat=False: dropped.clone())
    xb, x0b = sampler.p_sample_ddim(setup.x_T, setup.cond, ts, index=1, **setup.common)
    assert torch.equal(xa, xb)
    with pytest.raises(AttributeError):
        sampler.p_sample_ddim(setup.x_T, setup.cond, ts, index=1, quantize_denoised=True, **setup.common)
    setup.monkeypatch.setattr(setup.model.first_stage_model, "quantize", lambda z: (torch.round(z * 4) / 4, None, None),
                              raising=False)
    xq, x0q = sampler.p_sample_ddim(setup.x_T, setup.cond, ts, index=1, quantize_denoised=True, **setup.common)
    assert torch.equal(x0q, torch.round(x0b * 4) / 4)
    coef = sampler.step_coefficients(1, setup.s["cfg_scale"], setup.s["guidance_rescale"])
    assert rel_l2(xq, xb + coef[5] * (x0q - x0b)) < 1e-6
