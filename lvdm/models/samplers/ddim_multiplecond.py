"""DDIM sampler with three-way classifier-free guidance (reference: lvdm/models/samplers/ddim_multiplecond.py,
p_sample_ddim 210-279): v = e_uncond + cfg_img (e_img - e_uncond) + s (e_cond - e_img), where e_img is the pass
conditioned on the image tokens with an empty prompt (kwargs["unconditional_conditioning_img_nonetext"]).  Everything
else is the two-way sampler: same schedule code and options, same fused HIP update (mudg_ddim_step with its e_m input),
and the three UNet passes run as one tripled batch."""
from lvdm.models.samplers.ddim import DDIMSampler as _TwoWaySampler


class DDIMSampler(_TwoWaySampler):
    def _model_outputs(self, x, t, c, unconditional_conditioning, unconditional_guidance_scale, kwargs):
        kwargs = dict(kwargs)
        cfg_img = kwargs.pop("cfg_img", None)             # a named argument of the reference's p_sample_ddim, not a UNet input
        if cfg_img is None:
            cfg_img = unconditional_guidance_scale
        uc_img = kwargs["unconditional_conditioning_img_nonetext"]
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            return self.model.apply_model(x, t, c, **kwargs), None, None, 0.0
        if uc_img is None:
            raise ValueError("three-way guidance needs unconditional_conditioning_img_nonetext")
        conds = [c, unconditional_conditioning, uc_img]
        trio = self._batched_passes(x, t, conds, kwargs) if self.batch_cfg else None
        if trio is None:
            trio = tuple(self.model.apply_model(x, t, cd, **kwargs) for cd in conds)
        return trio[0], trio[1], trio[2], float(cfg_img)
