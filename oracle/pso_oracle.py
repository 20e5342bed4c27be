"""CPU oracle for the PSO latent-discovery hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the algorithm of the reference's hot path so that the
HIP kernels can be checked against it.  It must never be imported by the product
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may use it, and only as the checker / baseline.

Parity status: PINNED.  ``tests/golden/*.npz`` hold outputs of the reference's own
classes (``Swarm``, ``Discovery``, ``Generator``, ``Resnet``, ``Encoder``) run in the build
container by ``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py`` checks this
file against every one of them.

The arithmetic of the reference lives in PyTorch (a third-party dependency, torch 1.11
in its docker image, reference readme_cnr.txt:8): the network layers are restated here
with the same ``torch.nn.functional`` operators on CPU fp32 tensors; the swarm logic is
numpy float64 as in the reference.  Each function cites the reference lines it follows.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, used by every BN on the path


# --------------------------------------------------------------------------- networks
def _bn_eval(x, sd, prefix):
    """BatchNorm2d in eval mode (running statistics)."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS)


def generator_forward(sd: dict, z: torch.Tensor, stages: bool = False):
    """z [B,Z,1,1] fp32 -> img [B,C,H,W] in [-1,1].

    Follows reference src/utils/util_dcgan.py:133-149 (ConvT k7 s1 p0 + BN + ReLU,
    ConvT k4 s2 p1 + BN + ReLU, ConvT k4 s2 p1, Tanh).  The first kernel size is read
    from the weights (7 in the reference; 16 for the 64x64 extension).
    """
    a1 = F.relu(_bn_eval(F.conv_transpose2d(z, sd["gen.0.0.weight"], sd["gen.0.0.bias"], 1, 0), sd, "gen.0.1"))
    a2 = F.relu(_bn_eval(F.conv_transpose2d(a1, sd["gen.1.0.weight"], sd["gen.1.0.bias"], 2, 1), sd, "gen.1.1"))
    img = torch.tanh(F.conv_transpose2d(a2, sd["gen.2.weight"], sd["gen.2.bias"], 2, 1))
    if stages:
        return img, (a1, a2)
    return img


def rescale_per_image(img: torch.Tensor) -> torch.Tensor:
    """(img - min) / (max - min) with min/max PER IMAGE.

    Reference src/utils/util_data.py:129-135 takes the min/max of the whole tensor it is
    given; the reference only ever hands it a batch of one (util_discovery.py:48), so the
    batched equivalent is a per-image reduction.
    """
    b = img.shape[0]
    flat = img.reshape(b, -1)
    mn = flat.min(dim=1).values.reshape(b, 1, 1, 1)
    mx = flat.max(dim=1).values.reshape(b, 1, 1, 1)
    return (img - mn) / (mx - mn)


def resnet_depths(sd: dict):
    """Blocks per stage, read off the state-dict keys."""
    depths = []
    for li in range(1, 5):
        n = 0
        while f"layer{li}.{n}.conv1.weight" in sd:
            n += 1
        depths.append(n)
    return tuple(depths)


def resnet_forward(sd: dict, x: torch.Tensor, stages: bool = False):
    """x [B,C,H,W] -> logits [B,n_class].

    Follows reference src/pso/util_cnn.py:102-117 (stem, 4 stages, AdaptiveMaxPool2d,
    fc) and :173-190 (bottleneck: 1x1, 3x3 carrying the stride, 1x1, BN after each,
    projection shortcut on the first block of every stage, add, ReLU).
    """
    keep = []
    x = F.conv2d(x, sd["conv1.weight"], None, 2, 3)
    x = F.relu(_bn_eval(x, sd, "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    keep.append(x)
    for li, nblk in enumerate(resnet_depths(sd), start=1):
        for b in range(nblk):
            p = f"layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            idn = x
            y = F.relu(_bn_eval(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
            y = F.relu(_bn_eval(F.conv2d(y, sd[p + ".conv2.weight"], None, stride, 1), sd, p + ".bn2"))
            y = _bn_eval(F.conv2d(y, sd[p + ".conv3.weight"]), sd, p + ".bn3")
            if (p + ".identity_downsample.0.weight") in sd:
                idn = _bn_eval(F.conv2d(idn, sd[p + ".identity_downsample.0.weight"], None, stride),
                               sd, p + ".identity_downsample.1")
            x = F.relu(y + idn)
        keep.append(x)
    x = F.adaptive_max_pool2d(x, (1, 1)).reshape(x.shape[0], -1)  # named "avgpool" in the reference, but a MAX pool
    logits = F.linear(x, sd["fc.weight"], sd["fc.bias"])
    if stages:
        return logits, keep
    return logits


def alexnet_forward(sd: dict, x: torch.Tensor, padding: str = "same", activation: str = "LeakyReLU") -> torch.Tensor:
    """x [B,1,H,W] -> logits [B,n_class].  Follows reference src/pso/util_cnn.py:216-249: four times
    (convolution with bias, activation, MaxPool2d(2)), flatten in (channel, y, x) order, fc1 + activation,
    fc2 + activation, fc3; Dropout is inactive in eval mode.  ``activation``: "LeakyReLU" (slope 0.2,
    :48-50) or "ReLU"."""
    act = (lambda v: F.leaky_relu(v, 0.2)) if activation == "LeakyReLU" else F.relu
    for i in (1, 2, 3, 4):
        x = F.max_pool2d(act(F.conv2d(x, sd[f"conv{i}.weight"], sd[f"conv{i}.bias"], padding=padding)), 2)
    x = x.contiguous().view(x.shape[0], -1)
    x = act(F.linear(x, sd["fc1.weight"], sd["fc1.bias"]))
    x = act(F.linear(x, sd["fc2.weight"], sd["fc2.bias"]))
    return F.linear(x, sd["fc3.weight"], sd["fc3.bias"])


def encoder_forward(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """img [B,C,28,28] -> latent [B,Z,1,1].  Reference util_inverter.py:168-184."""
    x = F.leaky_relu(F.conv2d(x, sd["enc.0.weight"], sd["enc.0.bias"], 2, 1), 0.2)
    x = F.leaky_relu(F.conv2d(x, sd["enc.2.0.weight"], sd["enc.2.0.bias"], 2, 1), 0.2)
    return F.conv2d(x, sd["enc.3.weight"], sd["enc.3.bias"], 2, 0)


def vq_distances(inputs: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
    """[n, D] x [K, D] -> squared distances [n, K] as (|c|^2 + |x|^2) - 2 x.c^T in fp32
    (reference src/inverter/utils_vq_vae/util_function.py:12-17)."""
    flat = inputs.reshape(-1, codebook.shape[1])
    csq = torch.sum(codebook ** 2, dim=1)
    xsq = torch.sum(flat ** 2, dim=1, keepdim=True)
    return torch.addmm(csq + xsq, flat, codebook.t(), alpha=-2.0, beta=1.0)


def vq_nearest(inputs: torch.Tensor, codebook: torch.Tensor):
    """Nearest code per row of ``inputs`` [..., D]: (indices [...], codes [..., D]); the first
    of equal minima wins (reference util_function.py:19-20 and :34-44)."""
    d = vq_distances(inputs, codebook)
    idx = torch.min(d, dim=1)[1]
    codes = torch.index_select(codebook, 0, idx).view_as(inputs)
    return idx.view(*inputs.shape[:-1]), codes


def seed_codebook(last_positions) -> np.ndarray:
    """Codebook rows from the last-iteration positions of the swarms, one float64 [N, D] block per
    class in class order: cast to float32 and stacked (reference src/training/vq_vae.py:42-56,
    then util_model.py:49-54)."""
    return np.concatenate([np.asarray(p, dtype=np.float64).astype(np.float32) for p in last_positions], axis=0)


def vqvae_gan_encoder_forward(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """Encoder of VectorQuantizedVAE_GAN (reference util_model.py:273-279): the inverter encoder
    with an eval-mode BatchNorm before the second LeakyReLU."""
    x = F.leaky_relu(F.conv2d(x, sd["encoder.0.weight"], sd["encoder.0.bias"], 2, 1), 0.2)
    x = F.conv2d(x, sd["encoder.2.0.weight"], sd["encoder.2.0.bias"], 2, 1)
    x = F.leaky_relu(_bn_eval(x, sd, "encoder.2.1"), 0.2)
    return F.conv2d(x, sd["encoder.3.weight"], sd["encoder.3.bias"], 2, 0)


def vqvae_gan_forward(sd: dict, codebook: torch.Tensor, x: torch.Tensor):
    """(x_tilde, z_e_x, z_q_x, latents) of VectorQuantizedVAE_GAN.forward / encode (reference
    util_model.py:308-322), inference values only."""
    z_e_x = vqvae_gan_encoder_forward(sd, x)
    nhwc = z_e_x.permute(0, 2, 3, 1).contiguous()
    latents, codes = vq_nearest(nhwc, codebook)
    z_q_x = codes.permute(0, 3, 1, 2).contiguous()
    gsd = {"gen." + k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    return generator_forward(gsd, z_q_x), z_e_x, z_q_x, latents


def fitness_from_logits(logits: torch.Tensor, class_index: int, mode: str, threshold: float,
                        eps: float) -> torch.Tensor:
    """logits [B,n_class] -> fitness [B] fp32.

    Reference src/pso/util_discovery.py:69-80: softmax over all classes in fp32, pick the
    class (index 1 when there are only two), p' = min(p + threshold, 1), then p' + eps
    ("optimize_in_training") or (1 - p') + eps ("optimize_out_training").
    """
    p = F.softmax(logits, dim=1, dtype=torch.float32)
    p = p[:, class_index] if logits.shape[1] > 2 else p[:, 1]
    pc = torch.min(p + threshold, torch.ones(1))
    if mode == "optimize_in_training":
        return pc + eps
    if mode == "optimize_out_training":
        return (1.0 - pc) + eps
    raise ValueError(mode)


class Fitness:
    """The fitness function of one swarm: latent -> generator -> rescale -> CNN -> scalar.

    Restates reference util_discovery.py:33-101.  ``targets`` (one image per particle,
    fp32 in [-1,1]) switches on the inverter variant, where the pixel MSE against the
    un-rescaled generator output is added and eps is counted twice (:84-101).
    """

    def __init__(self, gen_sd, cnn_sd, class_index, mode="optimize_out_training", threshold=0.0,
                 eps=0.1, targets=None, w_ass=1.0, w_rec=1.0, cnn_args=None):
        self.gen_sd, self.cnn_sd = gen_sd, cnn_sd
        self.cnn_args = cnn_args  # given: the assessor is the "AlexNet" (padding, cnn_activation)
        self.class_index, self.mode = class_index, mode
        self.threshold, self.eps = threshold, eps
        self.targets, self.w_ass, self.w_rec = targets, w_ass, w_rec

    def images(self, pos: np.ndarray):
        """pos [B,D] float64 -> (img01, img), the fp64->fp32 cast of util_discovery.py:41-43 included."""
        z = torch.from_numpy(np.ascontiguousarray(pos)).to(torch.float32).reshape(pos.shape[0], -1, 1, 1)
        img = generator_forward(self.gen_sd, z)
        return rescale_per_image(img), img

    def __call__(self, pos: np.ndarray, first: int = 0) -> np.ndarray:
        """pos [B,D] -> fitness [B] float32 (particles ``first`` .. ``first+B-1`` of the swarm)."""
        with torch.no_grad():
            img01, img = self.images(pos)
            if self.cnn_args is not None:
                logits = alexnet_forward(self.cnn_sd, img01, self.cnn_args["padding"], self.cnn_args["cnn_activation"])
            else:
                logits = resnet_forward(self.cnn_sd, img01)
            f = fitness_from_logits(logits, self.class_index, self.mode, self.threshold, self.eps)
            f = f.numpy().astype(np.float32)
            if self.targets is not None:
                t = self.targets[first:first + pos.shape[0]]
                # util_discovery.py:97: mean over all pixels of one image, fp32
                mse = ((t.float() - img.float()) ** 2).reshape(pos.shape[0], -1).mean(dim=1).numpy()
                # :87,:98-101: python-float weights times float32 arrays stay float32
                f = (np.float32(self.w_ass) * f + np.float32(self.w_rec) * mse.astype(np.float32))
                f = (f + np.float32(self.eps)).astype(np.float32)
        return f


# ------------------------------------------------------------------------------ swarm
def init_swarm(n: int, d: int, rng=np.random):
    """Initial positions and velocities, consuming the RNG exactly like N successive
    ``Particle(dim)`` constructions (reference util_pso.py:30-31, 88-91): per particle
    ``randn(d)`` for the position then ``randn(d)`` for the velocity."""
    r = rng.randn(n, 2, d)
    return r[:, 0, :].copy(), (r[:, 1, :] - 0.5) / 10


def init_velocities(n: int, d: int, rng=np.random):
    """Velocities only (inverter variant: positions come from the encoder; the
    constructor then draws one ``randn(d)`` per particle, util_pso.py:26-31)."""
    return (rng.randn(n, d) - 0.5) / 10


def mean_pairwise_distance(pos: np.ndarray) -> float:
    """Mean Euclidean distance over all unordered pairs (reference util_pso.py:76-86,
    where it is called ``mse``)."""
    n = pos.shape[0]
    iu = np.triu_indices(n, 1)
    diff = pos[iu[0]] - pos[iu[1]]
    return float(np.mean(np.sqrt(np.sum(diff * diff, axis=1))))


def pairwise_distance_part(pos: np.ndarray, part: int, n_parts: int) -> float:
    """Sum of the Euclidean distances of the pairs (i < j) with i % n_parts == part: one of n_parts
    pieces of the sum behind mean_pairwise_distance (reference util_pso.py:76-86; the pieces add up to
    mean * n (n - 1) / 2).  The reference computes the whole sum in one process; the pieces exist to check
    the multi-process loop, which shares the work between its ranks."""
    n = pos.shape[0]
    iu = np.triu_indices(n, 1)
    keep = iu[0] % n_parts == part
    diff = pos[iu[0][keep]] - pos[iu[1][keep]]
    return float(np.sum(np.sqrt(np.sum(diff * diff, axis=1))))


def run_swarm(fitness, pos, vel, n_iterations, w_inertia, w_cogn, w_soci, *, tolerance=1e-4,
              schedule_inertia=False, early_stopping=True, rand=None, rng=np.random,
              batch: int = 1, record=True, diagnostics=True, f32_first_move=False):
    """The optimisation loop of reference util_pso.py:167-193, on SoA arrays.

    ``fitness(pos[B,D], first) -> float32[B]``.  ``batch=1`` evaluates one particle per
    call like the reference; larger batches only change how many are pushed through the
    networks at once.  ``rand`` [T,N,2] supplies (r1, r2) per particle per iteration; when
    None they are drawn from ``rng`` in the reference's order (util_pso.py:45,47: r1 then
    r2, particle by particle).

    ``f32_first_move`` restates the inverter variant (util_pso.py:93-112): positions come
    out of the encoder as float32 arrays, so in the first move ``g_best_pos - pos`` and
    ``p_best_pos - pos`` are float32 differences multiplied by a python float (numpy keeps
    float32 for that), and only the sum with the float64 inertia term promotes to float64.
    From the second move on every position is float64.

    Returns a dict with the quantities the reference exposes: ``g_best_val`` (list that
    grows only on improvement, :143-149), ``g_best_val_dummy`` (one entry per iteration,
    :151), ``g_best_pos``, ``mean_mse`` (:182-183), ``i`` (:193) and, when ``record``,
    per-iteration snapshots.
    """
    pos = np.array(pos, dtype=np.float64)
    vel = np.array(vel, dtype=np.float64)
    n, d = pos.shape
    pbest_pos = pos.copy()
    pbest_val = np.full(n, np.inf, dtype=np.float32)
    g_best_val = [np.inf]
    g_best_val_dummy = []
    g_best_pos = np.zeros(d)
    g_best_idx = -1
    out = {"mean_mse": [], "fitness": [], "pos": [pos.copy()], "vel": [vel.copy()], "pbest_val": [],
           "gbest_idx": [], "gbest_pos": [], "margin": np.inf}
    w = float(w_inertia)
    i = 1
    while i < n_iterations + 1:
        # update_personal_best (:114-125)
        fit = np.empty(n, dtype=np.float32)
        for s in range(0, n, batch):
            fit[s:s + batch] = fitness(pos[s:s + batch], s)
        finite = np.isfinite(pbest_val)
        if finite.any():
            out["margin"] = min(out["margin"], float(np.min(np.abs(fit[finite] - pbest_val[finite]))))
        better = fit < pbest_val
        pbest_val = np.where(better, fit, pbest_val)
        pbest_pos = np.where(better[:, None], pos, pbest_pos)
        # update_global_best (:135-151): strict '<' scan in particle order
        exp_val, exp_idx = g_best_val[-1], -1
        for k in range(n):
            if pbest_val[k] < exp_val:
                exp_val, exp_idx = pbest_val[k], k
        if exp_val < g_best_val[-1]:
            if g_best_val[-1] == np.inf:
                g_best_val[-1] = exp_val
            else:
                g_best_val.append(exp_val)
            g_best_pos = pbest_pos[exp_idx].copy()
            g_best_idx = exp_idx
        g_best_val_dummy.append(exp_val)
        # inertia decay from iteration 2 on (:178-179, :72-74)
        if i > 1 and schedule_inertia:
            w = 0.99 * w
        # move_particles (:153-157, :43-50, :39-41)
        r = rand[i - 1] if rand is not None else rng.random((n, 2))
        r1, r2 = r[:, 0:1], r[:, 1:2]
        inertia = w * vel
        if f32_first_move and i == 1:
            p32 = pos.astype(np.float32)
            best_cogn = ((w_cogn * r1).astype(np.float32) * (g_best_pos.astype(np.float32)[None, :] - p32)).astype(np.float64)
            best_soci = ((w_soci * r2).astype(np.float32) * (pbest_pos.astype(np.float32) - p32)).astype(np.float64)
        else:
            best_cogn = (w_cogn * r1) * (g_best_pos[None, :] - pos)
            best_soci = (w_soci * r2) * (pbest_pos - pos)
        vel = inertia + best_soci + best_cogn
        pos = pos + vel
        if diagnostics:
            out["mean_mse"].append(mean_pairwise_distance(pos))
        if record:
            out["fitness"].append(fit.copy())
            out["pos"].append(pos.copy())
            out["vel"].append(vel.copy())
            out["pbest_val"].append(pbest_val.copy())
            out["gbest_idx"].append(g_best_idx)
            out["gbest_pos"].append(g_best_pos.copy())
        # early stop on the last two IMPROVEMENTS (:186-188); i is not incremented on break
        if i > 2 and len(g_best_val) > 2 and early_stopping:
            if abs(g_best_val[-1] - g_best_val[-2]) < tolerance:
                break
        i += 1
    out.update(g_best_val=g_best_val, g_best_val_dummy=g_best_val_dummy, g_best_pos=g_best_pos,
               i=i, pbest_pos=pbest_pos, w_inertia=w)
    return out
