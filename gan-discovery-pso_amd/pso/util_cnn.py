"""Assessor CNN factory with the reference's signature and state_dict layout
(reference src/pso/util_cnn.py:24-38, 81-190), forward pass in HIP.

Provided: the bottleneck ResNets ("ResNet50" / "ResNet101" / "ResNet152") and the small
"AlexNet" (reference util_cnn.py:193-249; like the reference's it cannot be constructed with
the YAML's padding "valid" at 28x28, where the 4th convolution would see a 1x1 input: both raise
a RuntimeError).  "VGG16" / "InceptionNet" return None in the reference (:27-30) and here.
Training loops are outside the PSO hot path.
"""
from __future__ import annotations

import torch

from .. import engine
from ..modules import ParamTree, bn_rows

DEPTHS = {"ResNet50": (3, 4, 6, 3), "ResNet101": (3, 4, 23, 3), "ResNet152": (3, 8, 36, 3)}


class Resnet(ParamTree):
    """x [B, 1, S, S] -> logits [B, n_class].  Head is AdaptiveMaxPool2d + Linear(2048, n_class)."""

    def __init__(self, layers, image_channels, iid_classes, n_class=2):
        if image_channels != 1:
            raise ValueError("only image_channels = 1 is supported on the HIP path")
        rows = [("conv1.weight", (64, image_channels, 7, 7), "weight")] + bn_rows("bn1", 64)
        cin = 64
        for li, (nblk, mid) in enumerate(zip(layers, (64, 128, 256, 512)), start=1):
            for b in range(nblk):
                p = f"layer{li}.{b}"
                rows += [(f"{p}.conv1.weight", (mid, cin, 1, 1), "weight")] + bn_rows(f"{p}.bn1", mid)
                rows += [(f"{p}.conv2.weight", (mid, mid, 3, 3), "weight")] + bn_rows(f"{p}.bn2", mid)
                rows += [(f"{p}.conv3.weight", (mid * 4, mid, 1, 1), "weight")] + bn_rows(f"{p}.bn3", mid * 4)
                if b == 0:
                    rows += [(f"{p}.identity_downsample.0.weight", (mid * 4, cin, 1, 1), "weight")]
                    rows += bn_rows(f"{p}.identity_downsample.1", mid * 4)
                cin = mid * 4
        rows += [("fc.weight", (n_class, 2048), "weight"), ("fc.bias", (n_class,), "zeros")]
        super().__init__(rows)
        self.layers_per_stage = tuple(layers)
        self.n_class = n_class
        self.class_to_idx = {c: i for i, c in enumerate(sorted(iid_classes))}
        self.idx_to_class = {i: c for c, i in self.class_to_idx.items()}

    def device_handle(self, side, device=None) -> engine.PackedAssessor:
        dev = torch.device(device) if device is not None else self.param_device()
        return self.packed(lambda m: engine.PackedAssessor(m, dev, img_side=side), extra=(str(dev), side))

    def forward(self, x):
        return _forward_logits(self.device_handle(x.shape[-1], x.device), x)


def _forward_logits(ass, x):
    n, side = x.shape[0], x.shape[-1]
    img = x.reshape(n, side, side).to(torch.float32).contiguous()
    logits = torch.empty(n, ass.n_class, dtype=torch.float32, device=x.device)
    ws = torch.empty(ass.workspace_bytes(n), dtype=torch.uint8, device=x.device)
    L = engine._lib
    with torch.cuda.device(x.device):
        L.check(L.lib().gdpso_assessor_forward(ass.h, L.stream_handle(x.device), n, img.data_ptr(), logits.data_ptr(), 0, 0,
                                               L.MODE_OUT_TRAINING, 0.0, 0.0, 0, 1.0, 1.0, ws.data_ptr(), ws.numel()))
    return logits


class AlexNet(ParamTree):
    """x [B, 1, S, S] -> logits [B, n_class]: four convolutions (1 -> 32 -> 64 -> 128 -> 256, each with
    the activation and MaxPool2d(2)), three Linear layers (reference util_cnn.py:193-249).
    ``cnn_args``: the reference's dict with 'kernel', 'padding' ('same' | 'valid') and
    'cnn_activation' ('LeakyReLU' | 'ReLU')."""

    def __init__(self, image_channels, iid_classes, n_class, img_size, cnn_args):
        if image_channels != 1:
            raise ValueError("only image_channels = 1 is supported on the HIP path")
        k, padding = int(cnn_args["kernel"]), cnn_args["padding"]
        if cnn_args["cnn_activation"] not in ("LeakyReLU", "ReLU"):
            raise ValueError(cnn_args["cnn_activation"])  # reference get_activation, util_cnn.py:54
        pad = (k - 1) // 2 if padding == "same" else 0
        h = img_size
        for _ in range(4):
            hc = h + 2 * pad - k + 1
            if hc < 1:  # what torch raises from the reference's constructor (its dummy forward pass, :207-209)
                raise RuntimeError(f"Calculated padded input size per channel: ({h + 2 * pad} x {h + 2 * pad}). Kernel size: ({k} x {k}). "
                                   "Kernel size can't be greater than actual input size")
            h = hc // 2
        torch.randn(img_size, img_size)  # the reference's constructor draws its dummy input from torch's global stream
        to_linear = 256 * h * h
        rows, cin = [], image_channels
        for i, cout in enumerate((32, 64, 128, 256), start=1):
            rows += [(f"conv{i}.weight", (cout, cin, k, k), "weight"), (f"conv{i}.bias", (cout,), "zeros")]
            cin = cout
        rows += [("fc1.weight", (256, to_linear), "weight"), ("fc1.bias", (256,), "zeros"),
                 ("fc2.weight", (256, 256), "weight"), ("fc2.bias", (256,), "zeros"),
                 ("fc3.weight", (n_class, 256), "weight"), ("fc3.bias", (n_class,), "zeros")]
        super().__init__(rows)
        self._to_linear = to_linear
        self.cnn_args = dict(cnn_args)
        self.n_class = n_class
        self.class_to_idx = {c: i for i, c in enumerate(sorted(iid_classes))}
        self.idx_to_class = {i: c for c, i in self.class_to_idx.items()}

    def device_handle(self, side, device=None) -> engine.PackedAssessor:
        dev = torch.device(device) if device is not None else self.param_device()
        return self.packed(lambda m: engine.PackedAssessor(m, dev, img_side=side, cnn_args=self.cnn_args), extra=(str(dev), side))

    def forward(self, x):
        return _forward_logits(self.device_handle(x.shape[-1], x.device), x)


def get_cnn(model_name, image_channels, iid_classes, n_class, img_dim=64, cnn_args=None):
    if model_name == "AlexNet":
        return AlexNet(image_channels, iid_classes, n_class, img_dim, cnn_args)
    if model_name in ("VGG16", "InceptionNet"):
        return None  # `pass` in the reference (util_cnn.py:27-30)
    if model_name in DEPTHS:
        return Resnet(DEPTHS[model_name], image_channels, iid_classes, n_class)
    raise ValueError(model_name)
