"""Assessor CNN factory with the reference's signature and state_dict layout
(reference src/pso/util_cnn.py:24-38, 81-190), forward pass in HIP.

Provided: the bottleneck ResNets ("ResNet50" / "ResNet101" / "ResNet152").  The
reference's "AlexNet" cannot be constructed with its own YAML (padding "valid" at 28x28
leaves the 4th convolution a 1x1 input) and "VGG16" / "InceptionNet" return None there;
they raise here.  Training loops are outside the PSO hot path.
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
        n, side = x.shape[0], x.shape[-1]
        ass = self.device_handle(side, x.device)
        img = x.reshape(n, side, side).to(torch.float32).contiguous()
        logits = torch.empty(n, ass.n_class, dtype=torch.float32, device=x.device)
        ws = torch.empty(ass.workspace_bytes(n), dtype=torch.uint8, device=x.device)
        L = engine._lib
        with torch.cuda.device(x.device):
            L.check(L.lib().gdpso_assessor_forward(ass.h, L.stream_handle(x.device), n, img.data_ptr(), logits.data_ptr(), 0, 0,
                                                   L.MODE_OUT_TRAINING, 0.0, 0.0, 0, 1.0, 1.0, ws.data_ptr(), ws.numel()))
        return logits


def get_cnn(model_name, image_channels, iid_classes, n_class, img_dim=64, cnn_args=None):
    if model_name in DEPTHS:
        return Resnet(DEPTHS[model_name], image_channels, iid_classes, n_class)
    if model_name in ("AlexNet", "VGG16", "InceptionNet"):
        raise NotImplementedError(f"{model_name}: not available on the HIP path (see the module docstring)")
    raise ValueError(model_name)
