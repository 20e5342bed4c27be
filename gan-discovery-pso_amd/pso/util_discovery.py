"""Fitness of a latent position: generator -> per-image rescale -> assessor posterior.

Same constructor, methods, return types and errors as reference
src/pso/util_discovery.py:12-106; the arithmetic runs in the HIP library.  ``model_gan``
and ``model_cnn`` may be this package's modules or any object whose ``state_dict()`` has
the reference's key layout (weights are frozen: they are repacked once, at first use).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import engine


class Discovery:
    def __init__(self, iid_class, model_gan, model_cnn, device, control_pso_fitness="optimize_out_training",
                 iid_classes=None, obj_fun_threshold=0.0):
        self.model_gan = model_gan
        self.model_cnn = model_cnn
        self.device = torch.device(device)
        self.threshold = obj_fun_threshold
        self.iid_class = iid_class
        self.control_pso_fitness = control_pso_fitness
        if iid_classes is not None:
            self.class_to_idx = {c: i for i, c in enumerate(sorted(iid_classes))}
            self.idx_to_class = {i: c for c, i in self.class_to_idx.items()}
        # knob that does not exist in the reference (which is fp32 throughout): "bf16" / "f16" keep the
        # networks' activations and convolution weights in that type on the device
        self.compute_dtype = "f32"
        self._evaluator = None
        self._evaluator_key = None

    # ---- device objects
    @staticmethod
    def _model_key(model):
        """Changes whenever a weight of ``model`` is replaced or written in place (load_state_dict,
        .to(), optimiser steps): storage address and version counter of every tensor; numpy arrays
        (plain dicts of arrays are accepted as models) by address."""
        sd = model if isinstance(model, dict) else model.state_dict(keep_vars=True)
        key = []
        for v in sd.values():
            if isinstance(v, torch.Tensor):
                key.append((v.data_ptr(), v._version, str(v.device)))
            else:
                key.append((id(v),))
        return tuple(key)

    def evaluator(self) -> engine.Evaluator:
        """The packed device copy of the two networks; rebuilt when either model's weights have
        changed since it was packed (the reference calls the live modules every time)."""
        key = (self._model_key(self.model_gan), self._model_key(self.model_cnn), str(self.device), self.compute_dtype)
        if self._evaluator is None or self._evaluator_key != key:
            gen = engine.PackedGenerator(self.model_gan, self.device, dtype=self.compute_dtype)
            ass = engine.PackedAssessor(self.model_cnn, self.device, img_side=gen.side, dtype=self.compute_dtype)
            self._evaluator = engine.Evaluator(gen, ass)
            self._evaluator_key = key
        return self._evaluator

    def share_evaluator_with(self, other: "Discovery"):
        """Adopts ``other``'s packed device copy when both wrap the very same model objects, device and
        precision (several Discovery objects that differ only in iid_class then pack the weights once)."""
        if (other.model_gan is self.model_gan and other.model_cnn is self.model_cnn and other.device == self.device
                and other.compute_dtype == self.compute_dtype):
            self._evaluator, self._evaluator_key = other.evaluator(), other._evaluator_key
            return True
        return False

    def fitness_spec(self, eps=0.1, w_ass=1.0, w_rec=1.0) -> engine.FitnessSpec:
        n_class = self.evaluator().ass.n_class
        # more than two classes: the posterior of this swarm's class; otherwise index 1
        index = self.class_to_idx[self.iid_class] if n_class > 2 else 1
        spec = engine.FitnessSpec(class_index=index, mode=self.control_pso_fitness, threshold=float(self.threshold),
                                  eps=float(eps), w_ass=float(w_ass), w_rec=float(w_rec))
        spec.mode_code()  # ValueError for an unknown control_pso_fitness
        return spec

    def _latents(self, dim_space, pos):
        if not isinstance(pos, torch.Tensor):
            pos = torch.from_numpy(np.asarray(pos))
        pos = pos.to(self.device)
        if pos.dtype not in (torch.float32, torch.float64):
            pos = pos.float()
        return pos.reshape(-1, dim_space).contiguous()

    # ---- reference interface (one particle per call)
    def particles_to_img(self, dim_space, pos):
        z = self._latents(dim_space, pos)
        img01, img, _ = self.evaluator().images(z)
        s = img.shape[-1]
        return img01.reshape(-1, 1, s, s).clone(), img.reshape(-1, 1, s, s).clone()

    def fitness(self, dim_space, pos, eps=0.1):
        z = self._latents(dim_space, pos)
        fit, img01, img = self.evaluator().fitness(z, self.fitness_spec(eps))
        assert torch.max(img01) <= 1.0
        assert torch.min(img01) >= 0.0
        s = img.shape[-1]
        fit = fit.cpu().numpy()
        engine.check_chains(self.device)  # (the copy above has synchronised)
        return fit, img01.reshape(-1, 1, s, s).clone(), img.reshape(-1, 1, s, s).clone()

    def fitness_pso_inverter(self, dim_space, pos, img, eps=0.1, w_ass=1.0, w_rec=1.0):
        z = self._latents(dim_space, pos)
        if not isinstance(img, torch.Tensor):
            img = torch.from_numpy(np.asarray(img))
        ev = self.evaluator()
        s = ev.gen.side
        target = img.to(self.device, torch.float32).reshape(-1, s, s).contiguous()
        fit, img01, img_rec = ev.fitness(z, self.fitness_spec(eps, w_ass, w_rec), target=target)
        fit = fit.cpu().numpy()
        engine.check_chains(self.device)
        return fit, img01.reshape(-1, 1, s, s).clone(), img_rec.reshape(-1, 1, s, s).clone()

    # ---- batched form used by the swarm loop
    def fitness_batch(self, pos, eps=0.1, target=None, w_ass=1.0, w_rec=1.0):
        """pos [N, D] -> fitness [N] float32 device tensor (no host synchronisation)."""
        ev = self.evaluator()
        z = self._latents(ev.gen.z_dim, pos)
        fit, _, _ = ev.fitness(z, self.fitness_spec(eps, w_ass, w_rec), target=target)
        return fit

    def load_gan(self):
        pass

    def load_cnn(self):
        pass
