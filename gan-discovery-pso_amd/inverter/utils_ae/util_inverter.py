"""Encoder (and the generator copy) of the inverter package with the reference's
constructors and state_dict layouts (reference src/inverter/utils_ae/util_inverter.py:
119-140 and 164-184), forward passes in HIP.  Encoder training and gradient-based
inversion are outside the PSO hot path."""
from __future__ import annotations

import torch

from ... import engine
from ...modules import ParamTree
from ...utils.util_dcgan import Generator  # noqa: F401  (identical architecture, same keys)


class Encoder(ParamTree):
    """img [B, 1, 28, 28] in [-1, 1] -> latent [B, enc_dim, 1, 1]."""

    def __init__(self, enc_dim, channels_img, features_e=64):
        if channels_img != 1:
            raise ValueError("only channels_img = 1 is supported on the HIP path")
        rows = [("enc.0.weight", (features_e, channels_img, 4, 4), "weight"), ("enc.0.bias", (features_e,), "zeros"),
                ("enc.2.0.weight", (features_e * 2, features_e, 4, 4), "weight"), ("enc.2.0.bias", (features_e * 2,), "zeros"),
                ("enc.3.weight", (enc_dim, features_e * 2, 7, 7), "weight"), ("enc.3.bias", (enc_dim,), "zeros")]
        super().__init__(rows)
        self.enc_dim = enc_dim

    def device_handle(self, device=None) -> engine.PackedEncoder:
        dev = torch.device(device) if device is not None else self.param_device()
        return self.packed(lambda m: engine.PackedEncoder(m, dev), extra=(str(dev),))

    def forward(self, x):
        return self.device_handle(x.device).forward(x)
