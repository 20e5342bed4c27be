"""Parameter containers that reproduce the reference modules' ``state_dict`` layouts.

The reference builds its networks from torch layers; here a network is a table of
(dotted key, shape, kind) rows registered on nested ``nn.Module`` containers, so that
``state_dict()`` / ``load_state_dict()`` use exactly the reference's keys, while the
forward pass runs in the HIP library.  The packed device copy is rebuilt whenever a
parameter changes (checked through the tensors' version counters).
"""
from __future__ import annotations

import torch
from torch import nn


class ParamTree(nn.Module):
    def __init__(self, rows, nodes=None):
        """``nodes``: optional {top-level name: nn.Module} used as containers instead of bare
        ``nn.Module()`` ones (e.g. callable sub-networks)."""
        super().__init__()
        for name, module in (nodes or {}).items():
            self.add_module(name, module)
        for key, shape, kind in rows:
            *path, leaf = key.split(".")
            node = self
            for part in path:
                if part not in node._modules:
                    node.add_module(part, nn.Module())
                node = node._modules[part]
            if kind == "weight":
                node.register_parameter(leaf, nn.Parameter(torch.randn(shape) * 0.02))
            elif kind == "zeros":
                node.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))
            elif kind == "ones":
                node.register_parameter(leaf, nn.Parameter(torch.ones(shape)))
            elif kind == "buf_zeros":
                node.register_buffer(leaf, torch.zeros(shape))
            elif kind == "buf_ones":
                node.register_buffer(leaf, torch.ones(shape))
            elif kind == "counter":
                node.register_buffer(leaf, torch.zeros((), dtype=torch.long))
            else:
                raise ValueError(kind)
        self._packed = None
        self._packed_key = None

    def _fingerprint(self, extra=()):
        sd = self.state_dict(keep_vars=True)
        return tuple((v.data_ptr(), v._version) for v in sd.values()) + tuple(extra)

    def packed(self, build, extra=()):
        """Device handle built by ``build(self)``; rebuilt when any tensor changed."""
        key = self._fingerprint(extra)
        if self._packed is None or self._packed_key != key:
            self._packed = build(self)
            self._packed_key = key
        return self._packed

    def param_device(self):
        return next(self.parameters()).device


def bn_rows(prefix, c):
    return [(f"{prefix}.weight", (c,), "ones"), (f"{prefix}.bias", (c,), "zeros"),
            (f"{prefix}.running_mean", (c,), "buf_zeros"), (f"{prefix}.running_var", (c,), "buf_ones"),
            (f"{prefix}.num_batches_tracked", (), "counter")]
