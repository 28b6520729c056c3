"""AutoencoderKL at the drop-in boundary (reference: lvdm/models/autoencoder.py:13-107).  `encode` and `decode` run
the encoder / decoder on the gfx950 kernels (mudg_amd.engine.vae)."""
import torch
import torch.nn as nn

from lvdm.basics import Conv2d
from lvdm.distributions import DiagonalGaussianDistribution  # noqa: F401
from lvdm.modules.networks.ae_modules import Decoder, Encoder
from utils.utils import instantiate_from_config


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig, embed_dim, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None, test=False, logdir=None, input_dim=4, test_args=None):
        super().__init__()
        ddconfig = dict(ddconfig)
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.loss = instantiate_from_config(lossconfig)
        assert ddconfig["double_z"]
        self.quant_conv = Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim, self.input_dim = embed_dim, input_dim
        if monitor is not None:
            self.monitor = monitor
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    @property
    def device(self):
        return next(self.parameters()).device

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        self.load_state_dict(sd, strict=False)

    @torch.no_grad()
    def encode(self, x, **kwargs):
        """x (N, 3, H, W) in [-1, 1] -> DiagonalGaussianDistribution over (N, embed_dim, H/8, W/8): encoder +
        quant_conv on the HIP kernels (reference autoencoder.py:97-102)."""
        from mudg_amd.engine import vae
        return DiagonalGaussianDistribution(vae.encode_moments(self, x))

    @torch.no_grad()
    def decode(self, z, **kwargs):
        """z (N, z_channels, h, w) -> (N, 3, 8h, 8w): post_quant_conv + Decoder on the HIP kernels."""
        from mudg_amd.engine import vae
        return vae.decode(self, z)
