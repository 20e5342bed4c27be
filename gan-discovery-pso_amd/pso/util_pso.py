"""Particle swarm over a generator's latent space.

Same constructor, ``optimize`` signature, return values and post-run attributes as
reference src/pso/util_pso.py:24-193.  The reference walks a Python list of ``Particle``
objects and evaluates them one by one; here the swarm lives in HBM as structure-of-arrays
state and every iteration (generator, rescale, assessor, fitness, personal/global best,
move, mean pairwise distance) is a sequence of HIP kernels enqueued without host
round-trips.  ``Particle`` objects are materialised from the device history afterwards.

numpy's global RNG is consumed exactly as the reference consumes it (initial positions
and velocities particle by particle, then r1, r2 per particle per iteration), so a
script seeded with ``np.random.seed`` sees the same stream state after ``optimize``.

When ``torch.distributed`` is initialised with more than one rank, the particles are
sharded in contiguous blocks over the ranks and the global best is exchanged with one
all-gather per iteration; every rank returns the full result.
"""
from __future__ import annotations

import os

import numpy as np
import pandas as pd
import torch

from .. import engine


class Particle:
    """One particle's view of the finished run (reference util_pso.py:24-50 state)."""

    def __init__(self, dim, pos=None, img=None, vel=None):
        self.pos = pos if pos is not None else np.random.randn(dim)
        if img is not None:
            self.img = img
        self.vel = vel if vel is not None else (np.random.randn(dim) - 0.5) / 10
        self.p_best_pos = self.pos
        self.p_best_val = np.inf
        self.history = [self.pos]
        self.history_vel = [self.vel]


def _f32_1(v):
    return np.array([v], dtype=np.float32)


class Swarm:
    def __init__(self, plot_training_dir, discovery, num_particles, n_iterations, dim_space, device, tolerance=0.0001,
                 w_inertia=0.5, w_cogn=0.8, w_soci=0.9):
        self.plot_training_dir = plot_training_dir
        self.w_inertia = w_inertia
        self.w_cogn = w_cogn  # multiplies the GLOBAL-best attraction (the reference's naming)
        self.w_soci = w_soci  # multiplies the PERSONAL-best attraction
        self.tolerance = tolerance
        self.n_iterations = n_iterations
        self.dim_space = dim_space
        self.num_particles = num_particles
        self.discovery = discovery
        self.device = torch.device(device)
        self.g_best_pos = np.empty(self.dim_space)
        self.g_best_val = [np.inf]
        self.g_best_val_dummy = []
        self.swarm = []
        # knobs that do not exist in the reference
        self.record_history = True      # keep particle.history / history_vel (needed by checkpoint())
        self.diagnostics = True         # mean pairwise distance per iteration (history['mean_mse'])
        self.record_fitness = False     # keep every iteration's fitness vector in results['fitness']
        self.device_rng_seed = None     # set to an int to draw r1, r2 on the device (Philox) instead of numpy
        self.verbose = True
        self.comm = None                # an engine.TorchComm (or an object with its methods): shard the particles over its ranks
        self.rng = None                 # a numpy RandomState to draw from instead of numpy's global stream
        self.results = None

    # ------------------------------------------------------------------ helpers
    def _comm(self):
        if self.comm is not None:  # set explicitly: also a group of one rank goes through the sharded loop
            return self.comm
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return engine.TorchComm()
        return None

    def _encode_targets(self, ood_patient, encoder):
        """Initial positions from the encoder (reference util_pso.py:93-112): the first
        num_particles images of a batch-1 loader."""
        imgs = []
        for img_idx, (img, _) in enumerate(ood_patient):
            if img_idx >= self.num_particles:
                break
            assert img.dtype == torch.float32
            assert torch.max(img) <= 1.0
            assert torch.min(img) >= -1.0
            assert img[0].shape[0] == 1
            imgs.append(img[0:1])
        batch = torch.cat(imgs).to(self.device)
        if isinstance(encoder, engine.PackedEncoder):
            packed = encoder
        elif hasattr(encoder, "device_handle"):
            packed = encoder.device_handle(self.device)
        else:
            packed = engine.PackedEncoder(encoder, self.device)
        latents = packed.forward(batch).reshape(len(imgs), -1).cpu().numpy()  # float32, like the reference
        return latents, batch

    # ------------------------------------------------------------------ the loop
    def optimize(self, schedule_inertia=False, early_stopping=True, ood_patient=None, encoder=None):
        n, d, t = self.num_particles, self.dim_space, self.n_iterations
        rng = self.rng if self.rng is not None else np.random
        targets = None
        if ood_patient is not None:
            latents, targets = self._encode_targets(ood_patient, encoder)
            n = latents.shape[0]
            pos0 = latents.astype(np.float64)
            vel0 = (rng.randn(n, d) - 0.5) / 10
        else:
            r = rng.randn(n, 2, d)  # == N x (randn(d) for pos, randn(d) for vel)
            pos0, vel0 = r[:, 0, :].copy(), (r[:, 1, :] - 0.5) / 10
        rand = None
        if self.device_rng_seed is None:
            rng_state = rng.get_state()
            rand = rng.random_sample((t, n, 2))  # (r1, r2) per particle per iteration, reference order

        comm = self._comm()
        rank, world = (comm.rank, comm.world) if comm else (0, 1)
        lo, hi = engine.shard_bounds(n, world, rank)
        ev = self.discovery.evaluator()
        spec = self.discovery.fitness_spec()
        # ranks of equal size: the positions travel behind the candidate records, one collective per iteration
        sw = engine.HipSwarm(ev, hi - lo, d, t, first_index=lo, n_ranks=world, rank=rank, n_global=n,
                             history=self.record_history, diagnostics=self.diagnostics, record_fitness=self.record_fitness,
                             table_positions=comm is not None and self.diagnostics and n % world == 0)
        sw.f32_first_move = ood_patient is not None
        sw.load(pos0[lo:hi], vel0[lo:hi], self.w_inertia, rand=None if rand is None else rand[:, lo:hi, :],
                target=None if targets is None else targets[lo:hi])
        engine.run_loop(sw, t, spec, self.w_cogn, self.w_soci, schedule_inertia=schedule_inertia,
                        early_stopping=early_stopping, tolerance=self.tolerance, diagnostics=self.diagnostics,
                        seed=self.device_rng_seed or 0, comm=comm)
        res = sw.results()
        if comm:
            for key, axis in (("pos", 0), ("vel", 0), ("pbest_pos", 0), ("pbest_val", 0), ("hist_pos", 1), ("hist_vel", 1),
                              ("fitness", 1)):
                if key in res:
                    res[key] = comm.gather_rows(res[key], axis, n)
        iters = res["iters"]
        if rand is not None and iters < t:
            # leave numpy's stream where the reference would have left it
            rng.set_state(rng_state)
            rng.random_sample((iters, n, 2))
        return self._finish(res, n, targets, sw.pos if comm is None else None)

    def _finish(self, res, n, targets, device_pos):
        """Publishes a finished run on this object and builds optimize()'s return value."""
        t, iters = self.n_iterations, res["iters"]
        self.results = res
        self._device_pos = device_pos
        self._publish(res, n, targets)

        history = {"mean_mse": [float(x) for x in res.get("mean_mse", [])], "global_best_val": []}
        if self.verbose:
            for it in range(iters):
                dist_txt = history["mean_mse"][it] if history["mean_mse"] else float("nan")
                print(f"Iteration: {it + 1}, mean euclidean distance: {dist_txt}, global best value: {_f32_1(res['g_best_val_dummy'][it])}")
        history_particles, history_particles_vel = self.checkpoint()
        i = iters if res["stopped"] else t + 1
        if self.verbose:
            print(f"The best position is: {self.g_best_pos} with value: {self.g_best_val[-1]}, in iteration number: {i}")
        return history, history_particles, history_particles_vel, i

    def _publish(self, res, n, targets):
        self.w_inertia = res["w_inertia"]
        vals = res["g_best_val"]
        self.g_best_val = [_f32_1(v) if np.isfinite(v) else np.inf for v in vals] or [np.inf]
        self.g_best_val_dummy = [_f32_1(v) if np.isfinite(v) else np.inf for v in res["g_best_val_dummy"]]
        if res["gbest_idx"] >= 0:
            self.g_best_pos = res["g_best_pos"]
        self.swarm = []
        for k in range(n):
            p = Particle(self.dim_space, pos=res["pos"][k], vel=res["vel"][k],
                         img=None if targets is None else targets[k:k + 1].cpu())
            p.p_best_pos = res["pbest_pos"][k]
            v = res["pbest_val"][k]
            p.p_best_val = _f32_1(v) if np.isfinite(v) else np.inf
            if "hist_pos" in res:
                p.history = list(res["hist_pos"][:, k, :])
                p.history_vel = list(res["hist_vel"][:, k, :])
            self.swarm.append(p)

    def final_positions(self):
        """Positions after the last move as a float64 [N, D] tensor on the device (not in the
        reference, which goes through pickled histories: training/vq_vae.py:42-48 takes the last
        history row of every particle; this is that row without leaving HBM)."""
        if self.results is None:
            raise RuntimeError("optimize() has not been run")
        if getattr(self, "_device_pos", None) is not None:
            return self._device_pos
        return torch.from_numpy(np.ascontiguousarray(self.results["pos"])).to(self.device)

    def checkpoint(self):
        """{particle_k: DataFrame(history)} and the same for velocities (reference util_pso.py:159-165)."""
        history_particles, history_particles_vel = {}, {}
        for p_idx, particle in enumerate(self.swarm):
            history_particles[f"particle_{p_idx}"] = pd.DataFrame(particle.history)
            history_particles_vel[f"particle_{p_idx}"] = pd.DataFrame(particle.history_vel)
        return history_particles, history_particles_vel

    def mse(self):
        """Mean Euclidean distance over all unordered particle pairs (reference util_pso.py:76-86)."""
        pos = np.stack([p.pos for p in self.swarm])
        iu = np.triu_indices(len(self.swarm), 1)
        return float(np.mean(np.linalg.norm(pos[iu[0]] - pos[iu[1]], axis=1)))

    def dump_images(self, iteration=None):
        """Optional stand-in for the reference's per-iteration PNG (util_pso.py:127-128):
        writes the current swarm's rescaled images as one .npy grid under plot_training_dir."""
        pos = np.stack([p.pos for p in self.swarm])
        ev = self.discovery.evaluator()
        img01, _, _ = ev.images(torch.from_numpy(pos).to(self.device))
        os.makedirs(self.plot_training_dir, exist_ok=True)
        path = os.path.join(self.plot_training_dir, f"pso_images_{iteration if iteration is not None else 'final'}.npy")
        np.save(path, img01.cpu().numpy())
        return path


def optimize_swarms(swarms, schedule_inertia=False, early_stopping=True):
    """``[sw.optimize(schedule_inertia, early_stopping) for sw in swarms]`` with every iteration of
    all the swarms evaluated as ONE batch through the generator and the assessor.

    Not in the reference, which builds one Swarm per iid class and runs them one after the other
    (src/training/pso_discovery.py:174-218: 8 classes x 32 particles); swarms of 32 particles leave the
    GPU idle between ~70 short launches per iteration, 8 of them in one 256-image batch cost about
    one iteration of one.  The swarms must share the generator, the assessor, device, fitness mode,
    threshold and dim_space (their ``discovery.iid_class`` is what differs).  Every swarm comes out
    as if it had run alone on the same draws.  numpy's global stream is consumed swarm by swarm in
    the reference's order (positions and velocities, then r1, r2 for every iteration); with early
    stopping a swarm that stops at iteration i still consumes the draws of all its n_iterations
    (the sequential reference would hand the unused ones to the next swarm), so to reproduce a
    sequential early-stopped reference run exactly, run the swarms sequentially."""
    if not swarms:
        return []
    first = swarms[0]
    ev = first.discovery.evaluator()
    specs, draws = [], []
    for sw in swarms:
        if sw.dim_space != first.dim_space or sw.device != first.device:
            raise ValueError("optimize_swarms: the swarms must share dim_space and device")
        if sw is not first and not sw.discovery.share_evaluator_with(first.discovery):
            raise ValueError("optimize_swarms: the swarms must share one generator, one assessor, device and precision")
        specs.append(sw.discovery.fitness_spec())
        n, d, t = sw.num_particles, sw.dim_space, sw.n_iterations
        r = np.random.randn(n, 2, d)
        rand = np.random.random((t, n, 2)) if sw.device_rng_seed is None else None
        draws.append((r[:, 0, :].copy(), (r[:, 1, :] - 0.5) / 10, rand))
    if len({(sp.mode, sp.threshold, sp.eps) for sp in specs}) != 1:
        raise ValueError("optimize_swarms: the swarms must share control_pso_fitness, threshold and eps")
    batch = engine.HipSwarmBatch(ev, [sw.num_particles for sw in swarms], first.dim_space, [sw.n_iterations for sw in swarms],
                                 history=[sw.record_history for sw in swarms], diagnostics=[sw.diagnostics for sw in swarms],
                                 record_fitness=[sw.record_fitness for sw in swarms])
    params = []
    for sw, hs, (pos0, vel0, rand) in zip(swarms, batch.swarms, draws):
        hs.load(pos0, vel0, sw.w_inertia, rand=rand)
        params.append(dict(w_cogn=sw.w_cogn, w_soci=sw.w_soci, schedule_inertia=schedule_inertia, early_stopping=early_stopping,
                           tolerance=sw.tolerance, diagnostics=sw.diagnostics, seed=sw.device_rng_seed or 0))
    batch.run(specs, params)
    return [sw._finish(hs.results(), sw.num_particles, None, hs.pos) for sw, hs in zip(swarms, batch.swarms)]
