"""Generator with the reference's constructor and state_dict layout
(reference src/utils/util_dcgan.py:128-149), forward pass in HIP.

Only the frozen-forward path is provided: GAN training, the discriminator and the
evaluation hooks of the reference module are outside the PSO hot path.
"""
from __future__ import annotations

import torch

from .. import engine
from ..modules import ParamTree, bn_rows


class Generator(ParamTree):
    """z [B, z_dim, 1, 1] -> image [B, 1, S, S] in [-1, 1]  (S = 28 in the reference).

    ``first_kernel`` (7 in the reference, giving 7 -> 14 -> 28) is an extension:
    16 gives the 64x64 variant 16 -> 32 -> 64, which the reference does not have.
    """

    def __init__(self, z_dim, channels_img, features_g, first_kernel=7):
        if channels_img != 1:
            raise ValueError("only channels_img = 1 is supported on the HIP path")
        c1, c2 = features_g * 2, features_g
        rows = [("gen.0.0.weight", (z_dim, c1, first_kernel, first_kernel), "weight"), ("gen.0.0.bias", (c1,), "zeros")]
        rows += bn_rows("gen.0.1", c1)
        rows += [("gen.1.0.weight", (c1, c2, 4, 4), "weight"), ("gen.1.0.bias", (c2,), "zeros")]
        rows += bn_rows("gen.1.1", c2)
        rows += [("gen.2.weight", (c2, channels_img, 4, 4), "weight"), ("gen.2.bias", (channels_img,), "zeros")]
        super().__init__(rows)
        self.z_dim, self.channels_img, self.features_g = z_dim, channels_img, features_g

    def device_handle(self, device=None) -> engine.PackedGenerator:
        dev = torch.device(device) if device is not None else self.param_device()
        return self.packed(lambda m: engine.PackedGenerator(m, dev), extra=(str(dev),))

    def forward(self, x):
        return self.device_handle(x.device).forward(x)


def load_gan(model_dir, z_dim, channel, units_gen, device):
    """reference src/utils/util_dcgan.py:29-34"""
    import os
    generator = Generator(z_dim=z_dim, channels_img=channel, features_g=units_gen)
    checkpoint_g = torch.load(os.path.join(model_dir, "checkpoint_g.tar"), map_location=device, weights_only=True)
    generator.load_state_dict(checkpoint_g["model_state_dict"])
    generator.to(device)
    return generator
