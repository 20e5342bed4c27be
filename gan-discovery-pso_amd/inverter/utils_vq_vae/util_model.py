"""VQ-VAE model pieces that sit right after the PSO run, with the reference's constructors
and ``state_dict`` layouts (reference src/inverter/utils_vq_vae/util_model.py): the
codebook seeded from particle positions (:49-54, :125-148), the generator copy (:57-78)
and the inference methods of ``VectorQuantizedVAE_GAN`` (:268-322).  Forward passes run in
the HIP library; VQ-VAE training, the plain VAE and the PixelCNN prior are outside the path.
"""
from __future__ import annotations

import weakref

import numpy as np
import torch
from torch import nn

from ... import engine
from ...modules import ParamTree, bn_rows
from ...utils.util_dcgan import Generator as _DcganGenerator
from . import util_function


def freeze_parameters(model):
    for p in model.parameters():
        p.requires_grad = False
    return model


@torch.no_grad()
def pso_weights(model, data):
    """Overwrite every nn.Embedding weight of ``model`` with the rows of ``data`` (a
    DataFrame of particle positions; anything with ``.values``, or an array/tensor).
    Reference util_model.py:49-54.  The weight stays on the device it was on."""
    values = data.values if hasattr(data, "values") and not isinstance(data, torch.Tensor) else data
    data_tensor = torch.as_tensor(np.asarray(values) if not isinstance(values, torch.Tensor) else values)
    for m in model.modules():
        if isinstance(m, nn.Embedding):
            m.weight.data = data_tensor.to(m.weight.device)


@torch.no_grad()
def pso_weights_from_positions(model, positions, row0=0):
    """Device-side form of the seeding: rows [row0, row0+n) of every embedding =
    float32(positions [n, D]), positions being float64 swarm state already in HBM (what
    vq_vae.py:35-57 reads back from pickled histories, without the round trip)."""
    for m in model.modules():
        if isinstance(m, nn.Embedding):
            engine.vq_seed_codebook(m.weight.data, positions, row0)


class Generator(_DcganGenerator):
    """Same 3-ConvTranspose stack and keys as the DCGAN generator (reference util_model.py:57-78)."""

    def __init__(self, embedded_dim, channels_img, features_g):
        super().__init__(embedded_dim, channels_img, features_g)


class VQEmbedding(nn.Module):
    """reference util_model.py:125-148"""

    def __init__(self, K, D, data_pso):
        super().__init__()
        self.embedding = nn.Embedding(K, D)
        if data_pso is not None:
            pso_weights(self.embedding, data_pso)
        else:
            self.embedding.weight.data.uniform_(-1. / K, 1. / K)

    def forward(self, z_e_x):
        """z_e_x [B, D, H, W] -> int64 code indices [B, H, W]"""
        z_e_x_ = z_e_x.permute(0, 2, 3, 1).contiguous()
        return util_function.vq(z_e_x_, self.embedding.weight)

    def straight_through(self, z_e_x):
        """-> (z_q_x, z_q_x_bar), both [B, D, H, W]: the selected codes (forward values of
        the straight-through estimator and of the codebook branch are the same numbers)."""
        z_e_x_ = z_e_x.permute(0, 2, 3, 1).contiguous()
        z_q_x_, indices = util_function.vq_st(z_e_x_, self.embedding.weight.detach())
        z_q_x = z_q_x_.permute(0, 3, 1, 2).contiguous()
        z_q_x_bar_ = engine.vq_lookup(self.embedding.weight, indices).view_as(z_e_x_)
        z_q_x_bar = z_q_x_bar_.permute(0, 3, 1, 2).contiguous()
        return z_q_x, z_q_x_bar


class _SubNet(nn.Module):
    """Container for one sub-network's parameters that is callable like the reference's
    ``nn.Sequential`` (``model.encoder(x)``, ``model.decoder(z)``)."""

    def __init__(self, which):
        super().__init__()
        self._which = which
        self._owner = None

    def forward(self, x):
        return self._owner()._handles(x.device)[self._which].forward(x)


class VectorQuantizedVAE_GAN(ParamTree):
    """Encoder -> codebook -> DCGAN-shaped decoder (reference util_model.py:268-322),
    inference only: ``encode``, ``decode`` and the forward values of ``forward``."""

    def __init__(self, channels_img, embedded_dim=64, num_embedding=512, data_pso=None, features_g=64, features_d=64):
        if channels_img != 1:
            raise ValueError("only channels_img = 1 is supported on the HIP path")
        fd, fg, z = features_d, features_g, embedded_dim
        rows = [("encoder.0.weight", (fd, channels_img, 4, 4), "weight"), ("encoder.0.bias", (fd,), "zeros"),
                ("encoder.2.0.weight", (fd * 2, fd, 4, 4), "weight"), ("encoder.2.0.bias", (fd * 2,), "zeros")]
        rows += bn_rows("encoder.2.1", fd * 2)
        rows += [("encoder.3.weight", (z, fd * 2, 7, 7), "weight"), ("encoder.3.bias", (z,), "zeros")]
        rows += [("decoder.0.0.weight", (z, fg * 2, 7, 7), "weight"), ("decoder.0.0.bias", (fg * 2,), "zeros")]
        rows += bn_rows("decoder.0.1", fg * 2)
        rows += [("decoder.1.0.weight", (fg * 2, fg, 4, 4), "weight"), ("decoder.1.0.bias", (fg,), "zeros")]
        rows += bn_rows("decoder.1.1", fg)
        rows += [("decoder.2.weight", (fg, channels_img, 4, 4), "weight"), ("decoder.2.bias", (channels_img,), "zeros")]
        super().__init__(rows, nodes={"encoder": _SubNet(0), "decoder": _SubNet(1)})
        self.encoder._owner = self.decoder._owner = weakref.ref(self)
        with torch.no_grad():  # the reference's weights_init: Xavier-uniform convolution weights, zero biases
            for key, p in self.named_parameters():
                if key.endswith(".weight") and p.dim() == 4:
                    nn.init.xavier_uniform_(p)
        self.codebook = VQEmbedding(K=num_embedding, D=embedded_dim, data_pso=data_pso)
        self.embedded_dim = embedded_dim
        self._enc = self._dec = None
        self._net_key = None

    def _handles(self, device):
        dev = torch.device(device)
        key = self._fingerprint((str(dev),))
        if self._net_key != key:
            sd = self.state_dict()
            self._enc = engine.PackedEncoder(sd, dev, prefix="encoder")
            self._dec = engine.PackedGenerator(sd, dev, prefix="decoder")
            self._net_key = key
        return self._enc, self._dec

    def encode(self, x):
        z_e_x = self.encoder(x)
        return self.codebook(z_e_x)

    def decode(self, latents):
        """latents int64 [B, 1, 1] -> image [B, 1, 28, 28]"""
        z_q_x = engine.vq_lookup(self.codebook.embedding.weight, latents).permute(0, 3, 1, 2)  # (B, D, H, W)
        return self.decoder(z_q_x.contiguous())

    def forward(self, x):
        z_e_x = self.encoder(x)
        z_q_x_st, z_q_x = self.codebook.straight_through(z_e_x)
        x_tilde = self.decoder(z_q_x_st)
        return x_tilde, z_e_x, z_q_x


def get_model(name, channels_img, embedded_dim, num_embedding, data_pso, num_hiddens=64, features_g=64, features_d=64):
    """reference util_model.py:23-31"""
    if name == 'vqvae_dcgan':
        return VectorQuantizedVAE_GAN(channels_img, embedded_dim, num_embedding, data_pso, features_g, features_d)
    if name in ('vqvae_mnist', 'vqvae'):
        raise NotImplementedError(f"{name}: only the 'vqvae_dcgan' model (the one whose decoder is the PSO generator) is provided")
    raise ValueError(name)
