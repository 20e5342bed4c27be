"""Nearest-code lookup with the reference's call signatures
(reference src/inverter/utils_vq_vae/util_function.py), computed by the HIP library.

Forward only: the reference wraps these in ``torch.autograd.Function`` for VQ-VAE training
(straight-through gradient, codebook gradient); training is outside the PSO path, so the
results carry no autograd graph.
"""
from __future__ import annotations

from ... import engine


def vq(inputs, codebook):
    """inputs [..., D], codebook [K, D] -> int64 indices [...] of the nearest code
    (reference util_function.py:6-23)."""
    return engine.vq_nearest(inputs, codebook)[0]


def vq_st(inputs, codebook):
    """-> (codes shaped like ``inputs``, flat int64 indices)  (reference util_function.py:34-44)."""
    idx, codes, _ = engine.vq_nearest(inputs, codebook, want_codes=True)
    return codes.view_as(inputs), idx.reshape(-1)


__all__ = ["vq", "vq_st"]
