"""Drop-in `lvdm` package: the reference's import paths, constructor kwargs, module tree and state_dict keys, with
every forward on the denoising path executed by hand-written gfx950 kernels (mudg_amd).  See INTEGRATION.md."""
