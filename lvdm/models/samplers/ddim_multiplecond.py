"""DDIM sampler with three-way classifier-free guidance (reference: lvdm/models/samplers/ddim_multiplecond.py,
p_sample_ddim 210-279): v = e_uncond + cfg_img (e_img - e_uncond) + s (e_cond - e_img), where e_img is the pass
conditioned on the image tokens with an empty prompt (kwargs["unconditional_conditioning_img_nonetext"]).  Everything
else is the two-way sampler: same schedule code, same fused HIP update (mudg_ddim_step with its e_m input), and the
three UNet passes run as one tripled batch."""
import torch

from lvdm.common import noise_like
from lvdm.models.samplers.ddim import DDIMSampler as _TwoWaySampler


class DDIMSampler(_TwoWaySampler):
    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, uc_type=None, cfg_img=None,
                      mask=None, x0=None, guidance_rescale=0.0, **kwargs):
        if use_original_steps or quantize_denoised or score_corrector is not None or noise_dropout > 0.:
            raise NotImplementedError("original-steps / quantised x0 / score corrector / noise dropout are not on the "
                                      "MuDG path")
        from mudg_amd import ops
        if cfg_img is None:
            cfg_img = unconditional_guidance_scale
        uc_img = kwargs["unconditional_conditioning_img_nonetext"]
        guided = unconditional_conditioning is not None and unconditional_guidance_scale != 1.
        e_u = e_m = None
        if not guided:
            e_c = self.model.apply_model(x, t, c, **kwargs)
        else:
            if uc_img is None:
                raise ValueError("three-way guidance needs unconditional_conditioning_img_nonetext")
            trio = self._batched_passes(x, t, [c, unconditional_conditioning, uc_img], kwargs) if self.batch_cfg else None
            if trio is not None:
                e_c, e_u, e_m = trio
            else:
                e_c = self.model.apply_model(x, t, c, **kwargs)
                e_u = self.model.apply_model(x, t, unconditional_conditioning, **kwargs)
                e_m = self.model.apply_model(x, t, uc_img, **kwargs)
        coef = self.step_coefficients(index, unconditional_guidance_scale if guided else 1.0,
                                      guidance_rescale if guided else 0.0, temperature) + [float(cfg_img)]
        # always drawn, as the reference does (ddim.py:272: sigma_t * noise_like(...)): with eta = 0 the term is 0 * noise, but
        # the device generator advances identically, so later draws under the same seed (n_samples > 1) match
        noise = noise_like(x.shape, x.device, repeat_noise)
        return ops.ddim_step(x, e_c, e_u, noise, coef, e_m=e_m)
