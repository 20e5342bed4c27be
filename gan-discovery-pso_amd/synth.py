"""Seeded synthetic weights for the frozen networks on the PSO path.

No trained checkpoint ships with the reference (its .gitignore excludes /models), so
parity fixtures and throughput runs use synthetic weights.  They are produced with
numpy's legacy ``RandomState`` (MT19937 + polar Box-Muller), whose stream is stable
across numpy versions, so the golden fixtures can be regenerated anywhere from a seed.

The dictionaries returned here use exactly the ``state_dict`` key layout of the
reference modules, so they load into either implementation:

* generator  -- reference ``src/utils/util_dcgan.py:128-149``
* ResNet     -- reference ``src/pso/util_cnn.py:81-190``
* encoder    -- reference ``src/inverter/utils_ae/util_inverter.py:164-184``
* AlexNet    -- reference ``src/pso/util_cnn.py:193-249``

Two styles:

``lively``  weights with He-scaled convolutions, randomised BatchNorm statistics and a
            scaled classifier, so that the assessor posterior actually varies across
            the swarm (DCGAN's N(0, 0.02) init gives a flat ~1e-5 landscape on which
            every compare is a near-tie and parity is meaningless).
``dcgan``   N(0, 0.02) convolutions and default BatchNorm statistics, the reference's own
            initialiser (``src/utils/util_dcgan.py:45-48``); fine for throughput.
"""
from __future__ import annotations

import numpy as np

RESNET_DEPTHS = {
    "ResNet50": (3, 4, 6, 3),
    "ResNet101": (3, 4, 23, 3),
    "ResNet152": (3, 8, 36, 3),
}


def _bn(rs: np.random.RandomState, c: int, prefix: str, out: dict, lively: bool,
        gamma_scale: float = 1.0) -> None:
    if lively:
        out[prefix + ".weight"] = (gamma_scale * rs.uniform(0.6, 1.4, c)).astype(np.float32)
        out[prefix + ".bias"] = (0.1 * rs.standard_normal(c)).astype(np.float32)
        out[prefix + ".running_mean"] = (0.1 * rs.standard_normal(c)).astype(np.float32)
        out[prefix + ".running_var"] = rs.uniform(0.6, 1.4, c).astype(np.float32)
    else:
        out[prefix + ".weight"] = (0.02 * rs.standard_normal(c)).astype(np.float32)
        out[prefix + ".bias"] = np.zeros(c, np.float32)
        out[prefix + ".running_mean"] = np.zeros(c, np.float32)
        out[prefix + ".running_var"] = np.ones(c, np.float32)
    out[prefix + ".num_batches_tracked"] = np.zeros((), np.int64)


def _normal(rs: np.random.RandomState, shape, std: float) -> np.ndarray:
    return (std * rs.standard_normal(shape)).astype(np.float32)


def generator_state(z_dim: int, channels_img: int = 1, features_g: int = 64, *, seed: int = 1,
                    style: str = "lively", first_kernel: int = 7) -> dict:
    """State dict of the 3-ConvTranspose generator (reference util_dcgan.py:133-139).

    ``first_kernel`` is 7 for the reference's 28x28 generator; 16 gives the 64x64
    extension (1x1 -> 16 -> 32 -> 64), which has no counterpart in the reference.
    """
    rs = np.random.RandomState(seed)
    lively = style == "lively"
    c1, c2 = features_g * 2, features_g
    k = first_kernel
    sd: dict = {}
    # ConvTranspose2d weight layout is [Cin, Cout, kH, kW].
    sd["gen.0.0.weight"] = _normal(rs, (z_dim, c1, k, k), (1.0 / np.sqrt(z_dim)) if lively else 0.02)
    sd["gen.0.0.bias"] = _normal(rs, (c1,), 0.1 if lively else 0.0)
    _bn(rs, c1, "gen.0.1", sd, lively)
    # each output pixel of a k4/s2 transposed conv sums 4 taps x Cin
    sd["gen.1.0.weight"] = _normal(rs, (c1, c2, 4, 4), np.sqrt(2.0 / (4 * c1)) if lively else 0.02)
    sd["gen.1.0.bias"] = _normal(rs, (c2,), 0.1 if lively else 0.0)
    _bn(rs, c2, "gen.1.1", sd, lively)
    sd["gen.2.weight"] = _normal(rs, (c2, channels_img, 4, 4), np.sqrt(1.0 / (4 * c2)) if lively else 0.02)
    sd["gen.2.bias"] = _normal(rs, (channels_img,), 0.1 if lively else 0.0)
    return sd


def resnet_state(model_name: str = "ResNet50", image_channels: int = 1, n_class: int = 8, *,
                 seed: int = 2, style: str = "lively", logit_gain: float = 4.0) -> dict:
    """State dict of the bottleneck ResNet assessor (reference util_cnn.py:81-190)."""
    rs = np.random.RandomState(seed)
    lively = style == "lively"
    depths = RESNET_DEPTHS[model_name]
    sd: dict = {}

    def conv(name, cout, cin, k):
        std = np.sqrt(2.0 / (cin * k * k)) if lively else 0.02
        sd[name + ".weight"] = _normal(rs, (cout, cin, k, k), std)

    conv("conv1", 64, image_channels, 7)
    _bn(rs, 64, "bn1", sd, lively)
    cin = 64
    for li, (nblk, mid) in enumerate(zip(depths, (64, 128, 256, 512)), start=1):
        for b in range(nblk):
            p = f"layer{li}.{b}"
            conv(p + ".conv1", mid, cin, 1)
            _bn(rs, mid, p + ".bn1", sd, lively)
            conv(p + ".conv2", mid, mid, 3)
            _bn(rs, mid, p + ".bn2", sd, lively)
            conv(p + ".conv3", mid * 4, mid, 1)
            # a damped last BN keeps the residual stream from growing 2x per block
            _bn(rs, mid * 4, p + ".bn3", sd, lively, gamma_scale=0.5)
            if b == 0:
                conv(p + ".identity_downsample.0", mid * 4, cin, 1)
                _bn(rs, mid * 4, p + ".identity_downsample.1", sd, lively, gamma_scale=0.7)
            cin = mid * 4
    if lively:
        sd["fc.weight"] = _normal(rs, (n_class, 2048), logit_gain / np.sqrt(2048.0))
        sd["fc.bias"] = _normal(rs, (n_class,), 0.1)
    else:
        bound = 1.0 / np.sqrt(2048.0)
        sd["fc.weight"] = rs.uniform(-bound, bound, (n_class, 2048)).astype(np.float32)
        sd["fc.bias"] = rs.uniform(-bound, bound, (n_class,)).astype(np.float32)
    return sd


def alexnet_to_linear(img_size: int, kernel: int = 3, padding: str = "same") -> int:
    """Flattened feature count of the 4-convolution "AlexNet" (reference util_cnn.py:207-209, 233-234)."""
    pad = (kernel - 1) // 2 if padding == "same" else 0
    h = img_size
    for _ in range(4):
        h = (h + 2 * pad - kernel + 1) // 2
    return 256 * h * h


def alexnet_state(n_class: int = 8, *, kernel: int = 3, img_size: int = 28, padding: str = "same", seed: int = 5,
                  logit_gain: float = 4.0) -> dict:
    """State dict of the small "AlexNet" assessor (reference util_cnn.py:197-213): conv1..conv4 with bias,
    fc1 (flattened -> 256), fc2 (256 -> 256), fc3 (256 -> n_class)."""
    rs = np.random.RandomState(seed)
    sd: dict = {}
    cin = 1
    for i, cout in enumerate((32, 64, 128, 256), start=1):
        sd[f"conv{i}.weight"] = _normal(rs, (cout, cin, kernel, kernel), np.sqrt(2.0 / (cin * kernel * kernel)))
        sd[f"conv{i}.bias"] = _normal(rs, (cout,), 0.1)
        cin = cout
    k = alexnet_to_linear(img_size, kernel, padding)
    sd["fc1.weight"] = _normal(rs, (256, k), np.sqrt(2.0 / k))
    sd["fc1.bias"] = _normal(rs, (256,), 0.1)
    sd["fc2.weight"] = _normal(rs, (256, 256), np.sqrt(2.0 / 256))
    sd["fc2.bias"] = _normal(rs, (256,), 0.1)
    sd["fc3.weight"] = _normal(rs, (n_class, 256), logit_gain / np.sqrt(256.0))
    sd["fc3.bias"] = _normal(rs, (n_class,), 0.1)
    return sd


def encoder_state(enc_dim: int, channels_img: int = 1, features_e: int = 64, *, seed: int = 3,
                  style: str = "lively") -> dict:
    """State dict of the inverter encoder (reference util_inverter.py:168-174)."""
    rs = np.random.RandomState(seed)
    lively = style == "lively"
    c1, c2 = features_e, features_e * 2
    sd: dict = {}
    sd["enc.0.weight"] = _normal(rs, (c1, channels_img, 4, 4), np.sqrt(2.0 / (16 * channels_img)) if lively else 0.02)
    sd["enc.0.bias"] = _normal(rs, (c1,), 0.1 if lively else 0.0)
    sd["enc.2.0.weight"] = _normal(rs, (c2, c1, 4, 4), np.sqrt(2.0 / (16 * c1)) if lively else 0.02)
    sd["enc.2.0.bias"] = _normal(rs, (c2,), 0.1 if lively else 0.0)
    sd["enc.3.weight"] = _normal(rs, (enc_dim, c2, 7, 7), np.sqrt(1.0 / (49 * c2)) if lively else 0.02)
    sd["enc.3.bias"] = _normal(rs, (enc_dim,), 0.1 if lively else 0.0)
    return sd


def vqvae_gan_state(embedded_dim: int, features_g: int = 64, features_d: int = 64, *, seed: int = 4) -> dict:
    """State dict of VectorQuantizedVAE_GAN without its codebook (reference
    src/inverter/utils_vq_vae/util_model.py:273-289): the inverter encoder with a BatchNorm
    after the second convolution under ``encoder.*`` and the generator stack under ``decoder.*``."""
    rs = np.random.RandomState(seed)
    sd: dict = {}
    for k, v in encoder_state(embedded_dim, 1, features_d, seed=seed + 1000).items():
        sd["encoder." + k[len("enc."):]] = v
    _bn(rs, features_d * 2, "encoder.2.1", sd, True)
    for k, v in generator_state(embedded_dim, 1, features_g, seed=seed + 2000).items():
        sd["decoder." + k[len("gen."):]] = v
    return sd


def to_torch(sd: dict):
    """numpy state dict -> dict of torch CPU tensors (same keys)."""
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def checksum(sd: dict) -> str:
    """Order-independent content hash of a state dict (pins a fixture's weights)."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()[:16]
