"""Device-resident swarm engine: owns the HIP handles and buffers and drives the loop.

Layering (top to bottom):

* ``pso/util_pso.Swarm`` / ``pso/util_discovery.Discovery`` -- the reference's Python
  interface (reference src/pso/util_pso.py, src/pso/util_discovery.py);
* this module -- weight extraction from ``state_dict`` layouts, buffer ownership, the
  single-GPU fused loop and the sharded multi-GPU loop (one collective per iteration);
* ``_lib`` -- ctypes binding of the C ABI (include/gdpso.h) of libgdpso.so, where all the
  arithmetic happens in HIP kernels.

PyTorch is used for device memory, streams and ``torch.distributed`` only.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

BN_FIELDS = ("weight", "bias", "running_mean", "running_var")


def _np(sd, key):
    v = sd[key]
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


# ------------------------------------------------------------------ state_dict readers
def generator_tensors(sd, prefix="gen"):
    """The 14 arrays gdpso_generator_create wants, from the generator's state_dict keys
    (reference src/utils/util_dcgan.py:133-139: gen.0.0, gen.0.1, gen.1.0, gen.1.1, gen.2;
    ``prefix="decoder"`` reads the same stack out of a VectorQuantizedVAE_GAN,
    src/inverter/utils_vq_vae/util_model.py:283-289)."""
    g = prefix
    keys = [f"{g}.0.0.weight", f"{g}.0.0.bias"] + [f"{g}.0.1.{f}" for f in BN_FIELDS]
    keys += [f"{g}.1.0.weight", f"{g}.1.0.bias"] + [f"{g}.1.1.{f}" for f in BN_FIELDS]
    keys += [f"{g}.2.weight", f"{g}.2.bias"]
    t = [_np(sd, k) for k in keys]
    z_dim, c1, k, _ = t[0].shape
    if t[12].shape[1] != 1:
        raise ValueError("only single-channel generators are supported (data.channel = 1)")
    return t, dict(z_dim=int(z_dim), features_g=int(c1 // 2), first_kernel=int(k))


def resnet_tensors(sd):
    """Arrays in the forward order gdpso_assessor_create_resnet wants, plus the depths and
    class count read off the keys (reference src/pso/util_cnn.py:81-190)."""
    t = [_np(sd, "conv1.weight")] + [_np(sd, f"bn1.{f}") for f in BN_FIELDS]
    if t[0].shape[1] != 1:
        raise ValueError("only single-channel assessors are supported (data.channel = 1)")
    depths = []
    for li in range(1, 5):
        b = 0
        while f"layer{li}.{b}.conv1.weight" in sd:
            p = f"layer{li}.{b}"
            for c in (1, 2, 3):
                t.append(_np(sd, f"{p}.conv{c}.weight"))
                t += [_np(sd, f"{p}.bn{c}.{f}") for f in BN_FIELDS]
            if b == 0:
                t.append(_np(sd, f"{p}.identity_downsample.0.weight"))
                t += [_np(sd, f"{p}.identity_downsample.1.{f}") for f in BN_FIELDS]
            b += 1
        depths.append(b)
    t += [_np(sd, "fc.weight"), _np(sd, "fc.bias")]
    return t, dict(depths=tuple(depths), n_class=int(t[-1].shape[0]))


def alexnet_tensors(sd):
    """The 14 arrays gdpso_assessor_create_alexnet wants (reference src/pso/util_cnn.py:197-213)."""
    names = ["conv1", "conv2", "conv3", "conv4", "fc1", "fc2", "fc3"]
    t = [_np(sd, f"{n}.{f}") for n in names for f in ("weight", "bias")]
    if t[0].shape[1] != 1:
        raise ValueError("only single-channel assessors are supported (data.channel = 1)")
    return t, dict(kernel=int(t[0].shape[2]), n_class=int(t[12].shape[0]), to_linear=int(t[8].shape[1]))


def alexnet_to_linear(img_side, kernel, padding):
    """Input width of the reference AlexNet's fc1 (src/pso/util_cnn.py:197-210: four convolutions with
    MaxPool2d(2) each, 256 channels at the end), or None where a convolution's input is smaller than
    its window."""
    pad = (kernel - 1) // 2 if padding == "same" else 0
    h = img_side
    for _ in range(4):
        hc = h + 2 * pad - kernel + 1
        if hc < 1:
            return None
        h = hc // 2
    return 256 * h * h if h >= 1 else None


def encoder_tensors(sd, prefix="enc"):
    """reference src/inverter/utils_ae/util_inverter.py:168-174: enc.0, enc.2.0, enc.3.
    ``prefix="encoder"`` reads the encoder of a VectorQuantizedVAE_GAN, which has a
    BatchNorm after the second convolution (src/inverter/utils_vq_vae/util_model.py:273-279)."""
    e = prefix
    keys = [f"{e}.0.weight", f"{e}.0.bias", f"{e}.2.0.weight", f"{e}.2.0.bias", f"{e}.3.weight", f"{e}.3.bias"]
    if f"{e}.2.1.weight" in sd:
        keys += [f"{e}.2.1.{f}" for f in BN_FIELDS]
    t = [_np(sd, k) for k in keys]
    return t, dict(enc_dim=int(t[4].shape[0]), features_e=int(t[0].shape[0]))


def _state_dict_of(model):
    return model if isinstance(model, dict) else model.state_dict()


# ---------------------------------------------------------------------------- handles
_chain_aborts_seen = {}  # device index -> chain aborts already reported there (the library counts per GPU)


def check_chains(device=None) -> None:
    """Raises if a chained convolution launch (several dependent layers in one persistent kernel) on
    ``device`` (default: the current one) gave up waiting for a step since the last call: its outputs,
    and everything computed from them, are wrong.  That happens only when the launch's workgroups
    cannot all be resident (another process' kernels holding compute units for seconds).
    Synchronises the device."""
    index = torch.cuda.current_device() if device is None else torch.device(device).index
    if index is None:
        index = torch.cuda.current_device()
    n, launches = C.c_uint(0), C.c_uint(0)
    with torch.cuda.device(index):
        check(_lib.lib().gdpso_chain_stats(C.byref(launches), C.byref(n)))
    if n.value != _chain_aborts_seen.get(index, 0):
        _chain_aborts_seen[index] = n.value
        raise _lib.GdpsoError("a chained convolution launch timed out waiting for its workgroups; results since the last check are invalid "
                              "(set GDPSO_CONV_CHAIN=0 to run one launch per layer)")


def plan_options(plan_images: int = 0, splitk: bool = True) -> None:
    """Options of the launch plans made from now on (gdpso_plan_options; pack the networks AFTER the
    call: a packed network keeps the plans it has made).  ``plan_images`` > 0 chooses every layer's
    tile shape as if each batch held that many images: the shape fixes the order of the fp32
    additions over K, so an image then gets the same bits in a batch of any size, and a swarm
    sharded over ranks reproduces the unsharded run bit for bit when both plan for the same size
    (e.g. the ranks' block).  0: every batch by its own size (fastest; results then agree across
    batch sizes to fp32 rounding only).  ``splitk`` False: never split K over workgroups."""
    check(_lib.lib().gdpso_plan_options(int(plan_images), int(bool(splitk))))


def resident_options(stages: int = 2, max_images: int = 0, min_images: int = 0) -> None:
    """Switches of the resident stages (gdpso_resident_options; pack the assessor AFTER the call).
    ``stages`` 0: stem and every ResNet stage through the tiled convolution kernels; 1: stem and first stage
    in one launch with their maps in LDS; 2 (default): the second stage too.  ``max_images`` / ``min_images``
    > 0: largest / smallest planned batch that takes the resident kernel (0: keep the current limit; -1: the
    default, which is no upper limit and 64 images below for fp32 networks, 1 with 16-bit storage).  The paths agree to fp32 rounding, not bit for bit."""
    check(_lib.lib().gdpso_resident_options(int(stages), int(max_images), int(min_images)))


def resident_generator(mode=1) -> None:
    """The generator of the reference's shape as one launch with its maps in LDS (gdpso_resident_generator_option).
    0 / False: the tiled kernels; 1 / True: where the one launch is expected to be faster (default); 2: wherever the
    batch limits of the resident launches allow.  From the next forward call."""
    check(_lib.lib().gdpso_resident_generator_option(int(mode)))


class _Handle:
    _destroy = None

    def __init__(self):
        self.h = C.c_void_p()

    def __del__(self):
        try:
            if self.h and self._destroy:
                getattr(_lib.lib(), self._destroy)(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


def _dtype_code(dtype) -> int:
    """"f32" (the reference's precision) | "bf16" | "f16": storage and matrix-instruction input type of
    a packed network (accumulation and epilogues are fp32 either way)."""
    if isinstance(dtype, torch.dtype):
        dtype = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}.get(dtype, str(dtype))
    if dtype not in _lib.DTYPES:
        raise ValueError(f"dtype must be one of {sorted(_lib.DTYPES)}, got {dtype!r}")
    return _lib.DTYPES[dtype]


def _device_guard(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.GdpsoError(f"the PSO hot path runs on an MI355X only (got device '{device}'); there is no CPU fallback")
    return device


class PackedGenerator(_Handle):
    """Generator weights repacked on the device (gdpso_generator_create)."""
    _destroy = "gdpso_generator_destroy"

    def __init__(self, model, device, prefix="gen", dtype="f32"):
        super().__init__()
        self.device = _device_guard(device)
        self.dtype = dtype
        tensors, self.meta = generator_tensors(_state_dict_of(model), prefix)
        arr, keep = _lib.host_float_pointers(tensors)
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_generator_create_typed(C.byref(self.h), self.meta["z_dim"], self.meta["features_g"],
                                                          self.meta["first_kernel"], arr, len(keep), _dtype_code(dtype)))
        self.z_dim = self.meta["z_dim"]
        self.side = _lib.lib().gdpso_generator_image_side(self.h)
        # nominal MACs x2: ConvT(z->2F,k) + ConvT(2F->F,k4s2: 4 taps/output) + ConvT(F->1)
        f, k = self.meta["features_g"], self.meta["first_kernel"]
        self.flops_per_image = 2.0 * (self.z_dim * 2 * f * k * k + (2 * k) ** 2 * f * 2 * f * 4 + (4 * k) ** 2 * f * 4)

    def workspace_bytes(self, n):
        return _lib.lib().gdpso_generator_workspace_bytes(self.h, n)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """z [n, z_dim(,1,1)] fp32 or fp64 on the device -> image [n, 1, S, S] in [-1, 1]."""
        n = z.shape[0]
        z = z.to(self.device).reshape(n, self.z_dim).contiguous()
        if z.dtype not in (torch.float32, torch.float64):
            z = z.float()
        img = torch.empty(n, self.side, self.side, dtype=torch.float32, device=self.device)
        img01 = torch.empty_like(img)
        ws = torch.empty(self.workspace_bytes(n), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_generator_forward(self.h, _lib.stream_handle(self.device), n, z.data_ptr(),
                                                     1 if z.dtype == torch.float64 else 0, img.data_ptr(), img01.data_ptr(),
                                                     0, 0, ws.data_ptr(), ws.numel()))
        return img.reshape(n, 1, self.side, self.side)


class PackedAssessor(_Handle):
    """Assessor weights repacked on the device: a bottleneck ResNet (gdpso_assessor_create_resnet_typed)
    or, when the state dict has the reference AlexNet's keys, that network
    (gdpso_assessor_create_alexnet; ``cnn_args`` = the reference's dict with 'padding' and
    'cnn_activation', default padding 'same' and LeakyReLU)."""
    _destroy = "gdpso_assessor_destroy"

    def __init__(self, model, device, img_side=28, dtype="f32", cnn_args=None):
        super().__init__()
        self.device = _device_guard(device)
        self.dtype = dtype
        sd = _state_dict_of(model)
        if cnn_args is None:
            cnn_args = getattr(model, "cnn_args", None)
        with torch.cuda.device(self.device):
            if "fc1.weight" in sd and "conv4.weight" in sd:
                tensors, self.meta = alexnet_tensors(sd)
                arr, keep = _lib.host_float_pointers(tensors)
                args = dict(cnn_args or {})
                if args.get("padding") is not None and args["padding"] not in ("same", "valid"):
                    raise ValueError(f"padding {args['padding']!r}: 'same' or 'valid'")
                # fc1's width fixes the geometry (reference util_cnn.py:207-210 derives it from a dummy
                # forward pass): the padding is checked against it, or read off it when not given
                fits = [p for p in ("same", "valid") if alexnet_to_linear(img_side, self.meta["kernel"], p) == self.meta["to_linear"]]
                if args.get("padding") is None:
                    if not fits:
                        raise ValueError(f"AlexNet state dict: fc1 takes {self.meta['to_linear']} features, which neither padding gives "
                                         f"for {img_side}x{img_side} images and kernel {self.meta['kernel']}")
                    args["padding"] = fits[0]
                elif args["padding"] not in fits:
                    raise ValueError(f"AlexNet state dict: fc1 takes {self.meta['to_linear']} features but padding {args['padding']!r} at "
                                     f"{img_side}x{img_side} with kernel {self.meta['kernel']} gives "
                                     f"{alexnet_to_linear(img_side, self.meta['kernel'], args['padding'])}")
                if args.get("cnn_activation") is None:
                    import warnings
                    warnings.warn("AlexNet state dict without cnn_args: the activation cannot be read off the weights, assuming the "
                                  "reference's default LeakyReLU (pass cnn_args={'cnn_activation': 'ReLU' | 'LeakyReLU', ...})")
                    args["cnn_activation"] = "LeakyReLU"
                act = {"ReLU": 1, "LeakyReLU": 2}.get(args["cnn_activation"])
                if act is None:
                    raise ValueError(args["cnn_activation"])  # the reference's get_activation error (util_cnn.py:54)
                check(_lib.lib().gdpso_assessor_create_alexnet(C.byref(self.h), self.meta["n_class"], img_side, self.meta["kernel"],
                                                               int(args["padding"] == "same"), act, arr, len(keep), _dtype_code(dtype)))
            else:
                tensors, self.meta = resnet_tensors(sd)
                arr, keep = _lib.host_float_pointers(tensors)
                depths = (C.c_int * 4)(*self.meta["depths"])
                check(_lib.lib().gdpso_assessor_create_resnet_typed(C.byref(self.h), depths, self.meta["n_class"], img_side, arr,
                                                                    len(keep), _dtype_code(dtype)))
        self.n_class = self.meta["n_class"]
        self.side = img_side
        self.flops_per_image = _lib.lib().gdpso_assessor_flops_per_image(self.h)

    def workspace_bytes(self, n):
        return _lib.lib().gdpso_assessor_workspace_bytes(self.h, n)


class PackedEncoder(_Handle):
    """Inverter encoder repacked on the device (gdpso_encoder_create)."""
    _destroy = "gdpso_encoder_destroy"

    def __init__(self, model, device, prefix="enc"):
        super().__init__()
        self.device = _device_guard(device)
        tensors, self.meta = encoder_tensors(_state_dict_of(model), prefix)
        arr, keep = _lib.host_float_pointers(tensors)
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_encoder_create(C.byref(self.h), self.meta["enc_dim"], self.meta["features_e"], arr, len(keep)))
        self.enc_dim = self.meta["enc_dim"]

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        """img [n,1,28,28] or [n,28,28] fp32 on the device -> latent [n, enc_dim, 1, 1]."""
        n = img.shape[0]
        x = img.to(self.device, torch.float32).reshape(n, 28, 28).contiguous()
        ws = torch.empty(_lib.lib().gdpso_encoder_workspace_bytes(self.h, n), dtype=torch.uint8, device=self.device)
        out = torch.empty(n, self.enc_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_encoder_forward(self.h, _lib.stream_handle(self.device), n, ptr(x), ptr(out), ptr(ws), ws.numel()))
        return out.reshape(n, self.enc_dim, 1, 1)


# ------------------------------------------------------------------ vector quantisation
def _vq_args(inputs: torch.Tensor, codebook: torch.Tensor):
    device = _device_guard(codebook.device)
    if codebook.dim() != 2:
        raise ValueError("codebook must be [K, D]")
    cb = codebook.detach().to(torch.float32).contiguous()
    return device, cb


def vq_nearest(inputs: torch.Tensor, codebook: torch.Tensor, want_codes=False, want_dist=False):
    """Nearest codebook row of every row of ``inputs`` [..., D] (gdpso_vq_nearest).
    Returns (indices int64 [...], codes [..., D] or None, best distance [...] or None)."""
    device, cb = _vq_args(inputs, codebook)
    k, d = cb.shape
    if inputs.shape[-1] != d:
        raise RuntimeError(f"shape '[-1, {d}]' is invalid for input of size {inputs.numel()}")  # the reference's view() error
    x = inputs.detach().to(device, torch.float32).reshape(-1, d).contiguous()
    n = x.shape[0]
    idx = torch.empty(n, dtype=torch.int64, device=device)
    codes = torch.empty(n, d, dtype=torch.float32, device=device) if want_codes else None
    dist = torch.empty(n, dtype=torch.float32, device=device) if want_dist else None
    ws = torch.empty(_lib.lib().gdpso_vq_workspace_bytes(k, d), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        check(_lib.lib().gdpso_vq_nearest(_lib.stream_handle(device), n, d, k, ptr(x), ptr(cb), ptr(idx), ptr(codes), ptr(dist),
                                          ptr(ws), ws.numel()))
    lead = inputs.shape[:-1]
    return (idx.reshape(lead), None if codes is None else codes.reshape(*lead, d), None if dist is None else dist.reshape(lead))


def vq_lookup(codebook: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    """codebook[indices] -> [..., D] (gdpso_vq_lookup); IndexError on an index outside the codebook."""
    device, cb = _vq_args(indices, codebook)
    k, d = cb.shape
    idx = indices.detach().to(device, torch.int64).reshape(-1).contiguous()
    n = idx.shape[0]
    if n and bool(((idx < 0) | (idx >= k)).any()):
        raise IndexError("index out of range in self")
    out = torch.empty(n, d, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        check(_lib.lib().gdpso_vq_lookup(_lib.stream_handle(device), n, d, k, ptr(cb), ptr(idx), ptr(out)))
    return out.reshape(*indices.shape, d)


def vq_seed_codebook(codebook: torch.Tensor, positions: torch.Tensor, row0: int = 0):
    """In place: codebook[row0 : row0+n] = float32(positions [n, D] float64 on the device)."""
    device = _device_guard(codebook.device)
    if codebook.dtype != torch.float32 or not codebook.is_contiguous():
        raise ValueError("codebook must be a contiguous float32 tensor")
    k, d = codebook.shape
    pos = positions.detach().to(device, torch.float64).reshape(-1, d).contiguous()
    with torch.cuda.device(device):
        check(_lib.lib().gdpso_vq_seed_codebook(_lib.stream_handle(device), ptr(codebook), k, d, int(row0), ptr(pos), pos.shape[0]))


# -------------------------------------------------------------------- fitness settings
@dataclass
class FitnessSpec:
    class_index: int = 1
    mode: str = "optimize_out_training"
    threshold: float = 0.0
    eps: float = 0.1
    w_ass: float = 1.0
    w_rec: float = 1.0

    def mode_code(self) -> int:
        if self.mode == "optimize_in_training":
            return _lib.MODE_IN_TRAINING
        if self.mode == "optimize_out_training":
            return _lib.MODE_OUT_TRAINING
        raise ValueError(self.mode)  # same error as reference util_discovery.py:80


class Evaluator:
    """latents -> images -> fitness on the device (generator + assessor handles + workspace)."""

    def __init__(self, gen: PackedGenerator, ass: PackedAssessor):
        if gen.side != ass.side:
            raise ValueError(f"generator makes {gen.side}x{gen.side} images but the assessor plan is for {ass.side}")
        self.gen, self.ass, self.device = gen, ass, gen.device
        self._ws = None
        self._img = {}

    def workspace(self, n):
        need = max(self.gen.workspace_bytes(n), self.ass.workspace_bytes(n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    MAX_IMAGE_BUFFER_SETS = 4  # distinct batch sizes whose image buffers are kept

    def image_buffers(self, n):
        if n not in self._img:
            while len(self._img) >= self.MAX_IMAGE_BUFFER_SETS:  # drop the set that was created first
                self._img.pop(next(iter(self._img)))
            s = self.gen.side
            self._img[n] = (torch.empty(n, s, s, dtype=torch.float32, device=self.device),
                            torch.empty(n, s, s, dtype=torch.float32, device=self.device),
                            torch.empty(n, dtype=torch.float32, device=self.device))
        return self._img[n]

    def images(self, z: torch.Tensor, target=None):
        """z [n,Z] float64/float32 device tensor -> (img01, img, mse|None), each [n,S,S]."""
        n = z.shape[0]
        img, img01, mse = self.image_buffers(n)
        ws = self.workspace(n)
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_generator_forward(self.gen.h, _lib.stream_handle(self.device), n, ptr(z),
                                                     1 if z.dtype == torch.float64 else 0, ptr(img), ptr(img01),
                                                     ptr(target), ptr(mse) if target is not None else 0, ptr(ws), ws.numel()))
        return img01, img, (mse if target is not None else None)

    def assess(self, img01: torch.Tensor, spec: FitnessSpec, mse=None, want_logits=False):
        n = img01.shape[0]
        ws = self.workspace(n)
        fit = torch.empty(n, dtype=torch.float32, device=self.device)
        logits = torch.empty(n, self.ass.n_class, dtype=torch.float32, device=self.device) if want_logits else None
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_assessor_forward(self.ass.h, _lib.stream_handle(self.device), n, ptr(img01), ptr(logits), ptr(fit),
                                                    spec.class_index, spec.mode_code(), spec.threshold, spec.eps, ptr(mse),
                                                    spec.w_ass, spec.w_rec, ptr(ws), ws.numel()))
        return fit, logits

    def fitness(self, z: torch.Tensor, spec: FitnessSpec, target=None):
        img01, img, mse = self.images(z, target)
        fit, _ = self.assess(img01, spec, mse)
        return fit, img01, img


# ------------------------------------------------------------------------ swarm state
class HipSwarm:
    """Swarm state in HBM (structure of arrays, float64 like the reference's numpy state)
    and the per-phase kernel launches.  One instance per process / GPU."""

    def __init__(self, evaluator: Evaluator, n_local, d, max_iter, *, first_index=0, n_ranks=1, rank=0, n_global=None,
                 history=True, diagnostics=True, record_fitness=False, pos=None, fitness=None, table_positions=False):
        """``pos`` [n, d] float64 / ``fitness`` [n] float32: views into buffers owned by someone else (a
        HipSwarmBatch lays its swarms' positions and fitness values out back to back).
        ``table_positions`` (ranks of equal size): this rank's positions live right behind its candidate
        record in the exchange table, so the one all-gather per iteration moves them too."""
        self.ev, self.device = evaluator, evaluator.device
        self.n, self.d, self.max_iter = int(n_local), int(d), int(max_iter)
        self.first_index, self.n_ranks, self.rank = int(first_index), int(n_ranks), int(rank)
        self.n_global = int(n_global if n_global is not None else n_local)
        if evaluator.gen.z_dim != self.d:
            raise ValueError(f"dim_space {d} != generator z_dim {evaluator.gen.z_dim}")
        dev, f64, f32 = self.device, torch.float64, torch.float32
        n, t = self.n, self.max_iter
        # the table the ranks all-gather once per iteration: per rank {value, global index, position[d],
        # distance sums[2]} and, with table_positions, the rank's particle positions behind that record
        cd = _lib.cand_doubles(d)
        self.table_positions = bool(table_positions)
        if self.table_positions:
            if pos is not None or self.n_global != n * self.n_ranks:
                raise ValueError("table_positions needs ranks of equal size and positions owned by the swarm")
            self.exchange_table = torch.zeros(self.n_ranks, cd + n * d, dtype=f64, device=dev)
            pos = self.exchange_table[self.rank, cd:].view(n, d)
        else:
            self.exchange_table = torch.zeros(self.n_ranks, cd, dtype=f64, device=dev)
        self.cand = self.exchange_table[:, :cd]
        self.pos = torch.zeros(n, d, dtype=f64, device=dev) if pos is None else pos
        self.vel = torch.zeros(n, d, dtype=f64, device=dev)
        self.pbest_pos = torch.zeros(n, d, dtype=f64, device=dev)
        self.pbest_val = torch.zeros(n, dtype=f32, device=dev)
        self.fitness = torch.zeros(n, dtype=f32, device=dev) if fitness is None else fitness
        assert self.pos.shape == (n, d) and self.pos.is_contiguous() and self.fitness.shape == (n,)
        self.gbest_pos = torch.zeros(d, dtype=f64, device=dev)
        self.ctrl = torch.zeros(_lib.CTRL_DOUBLES, dtype=f64, device=dev)
        self.g_best_val = torch.zeros(t + 1, dtype=f32, device=dev)
        self.g_best_val_dummy = torch.zeros(t, dtype=f32, device=dev)
        self.mean_dist = torch.zeros(t, dtype=f64, device=dev) if diagnostics else None
        self.hist_pos = torch.zeros(t + 1, n, d, dtype=f64, device=dev) if history else None
        self.hist_vel = torch.zeros(t + 1, n, d, dtype=f64, device=dev) if history else None
        nscr = max(self.n_global, n) if diagnostics else n
        self.scratch = torch.zeros(_lib.lib().gdpso_swarm_scratch_bytes(nscr), dtype=torch.uint8, device=dev)
        self.fit_hist = torch.zeros(t, n, dtype=f32, device=dev) if record_fitness else None
        self.rand = None
        self.target = None
        self.f32_first_move = False
        self.desc = _lib.SwarmDesc(
            n=n, d=d, first_index=self.first_index, n_ranks=self.n_ranks, pos=ptr(self.pos), vel=ptr(self.vel),
            pbest_pos=ptr(self.pbest_pos), pbest_val=ptr(self.pbest_val), fitness=ptr(self.fitness),
            gbest_pos=ptr(self.gbest_pos), ctrl=ptr(self.ctrl), g_best_val=ptr(self.g_best_val),
            g_best_val_dummy=ptr(self.g_best_val_dummy), mean_dist=ptr(self.mean_dist), hist_pos=ptr(self.hist_pos),
            hist_vel=ptr(self.hist_vel), cand=ptr(self.cand), scratch=ptr(self.scratch), scratch_bytes=self.scratch.numel(),
            max_iter=t, cand_stride=self.exchange_table.shape[1])

    # -- set-up
    def _stream(self):
        return _lib.stream_handle(self.device)

    def load(self, pos: np.ndarray, vel: np.ndarray, w_inertia: float, rand=None, target=None):
        self.pos.copy_(torch.from_numpy(np.ascontiguousarray(pos, dtype=np.float64)))
        self.vel.copy_(torch.from_numpy(np.ascontiguousarray(vel, dtype=np.float64)))
        self.rand = None if rand is None else torch.from_numpy(np.ascontiguousarray(rand, dtype=np.float64)).to(self.device)
        self.target = None if target is None else target.to(self.device, torch.float32).reshape(self.n, self.ev.gen.side, self.ev.gen.side).contiguous()
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_swarm_reset(C.byref(self.desc), self._stream(), float(w_inertia)))

    # -- phases (multi-process loop)
    def evaluate(self, i, spec: FitnessSpec):
        img, img01, mse = self.ev.image_buffers(self.n)
        ws = self.ev.workspace(self.n)
        L, s = _lib.lib(), self._stream()
        with torch.cuda.device(self.device):
            check(L.gdpso_generator_forward(self.ev.gen.h, s, self.n, ptr(self.pos), 1, ptr(img), ptr(img01), ptr(self.target),
                                            ptr(mse) if self.target is not None else 0, ptr(ws), ws.numel()))
            check(L.gdpso_assessor_forward(self.ev.ass.h, s, self.n, ptr(img01), 0, ptr(self.fitness), spec.class_index,
                                           spec.mode_code(), spec.threshold, spec.eps, ptr(mse) if self.target is not None else 0,
                                           spec.w_ass, spec.w_rec, ptr(ws), ws.numel()))
        if self.fit_hist is not None:
            self.fit_hist[i - 1].copy_(self.fitness)

    def local_best(self, i):
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_swarm_local_best(C.byref(self.desc), self._stream(), i, self.rank))

    def global_best(self, i, schedule_inertia, early_stopping, tolerance):
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_swarm_global_best(C.byref(self.desc), self._stream(), i, int(schedule_inertia),
                                                     int(early_stopping), float(tolerance)))

    def move(self, i, w_cogn, w_soci, seed=0):
        rnd = 0 if self.rand is None else self.rand[i - 1].data_ptr()
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_swarm_move(C.byref(self.desc), self._stream(), i, float(w_cogn), float(w_soci), rnd, int(seed),
                                              int(self.f32_first_move)))

    def mean_distance(self, i, pos_all: torch.Tensor):
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_swarm_mean_distance(C.byref(self.desc), self._stream(), i, ptr(pos_all), pos_all.shape[0]))

    def distance_part(self, i, pos_all: torch.Tensor):
        """This rank's share of the diagnostic of iteration i: the distances of every n_ranks-th tile of
        pairs, summed into its candidate record (one of two slots, by the iteration's parity)."""
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_swarm_distance_part(C.byref(self.desc), self._stream(), i, ptr(pos_all), pos_all.shape[0], self.rank))

    def distance_part_table(self, i):
        """distance_part on the positions that came with the records (table_positions)."""
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_swarm_distance_part_table(C.byref(self.desc), self._stream(), i, self.n_global, self.rank))

    def distance_finish(self, i):
        """The ranks' sums in the exchanged candidate table -> mean_dist[i - 1]."""
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_swarm_distance_finish(C.byref(self.desc), self._stream(), i, self.n_global))

    def after_exchange(self, i, schedule_inertia, early_stopping, tolerance, w_cogn, w_soci, seed=0, diag_iteration=0):
        """distance_finish(diag_iteration) (0: none), global_best(i) and move(i) as one call: one launch
        for swarms that one workgroup can step."""
        rnd = 0 if self.rand is None else self.rand[i - 1].data_ptr()
        with torch.cuda.device(self.device):
            check(_lib.lib().gdpso_swarm_after_exchange(C.byref(self.desc), self._stream(), i, int(schedule_inertia), int(early_stopping),
                                                        float(tolerance), float(w_cogn), float(w_soci), rnd, int(seed),
                                                        int(self.f32_first_move), int(diag_iteration), self.n_global))

    # -- fused loop (one process)
    def iterate(self, first, count, spec: FitnessSpec, w_cogn, w_soci, schedule_inertia, early_stopping, tolerance,
                diagnostics, seed=0):
        img, img01, mse = self.ev.image_buffers(self.n)
        ws = self.ev.workspace(self.n)
        run = _lib.RunDesc(class_index=spec.class_index, mode=spec.mode_code(), threshold=spec.threshold, eps=spec.eps,
                           target=ptr(self.target), w_ass=spec.w_ass, w_rec=spec.w_rec, w_cogn=w_cogn, w_soci=w_soci,
                           schedule_inertia=int(schedule_inertia), early_stopping=int(early_stopping), tolerance=tolerance,
                           rand=ptr(self.rand), seed=int(seed), diagnostics=int(bool(diagnostics) and self.mean_dist is not None),
                           f32_first_move=int(self.f32_first_move),
                           img01=ptr(img01), img=ptr(img), mse=ptr(mse), workspace=ptr(ws), workspace_bytes=ws.numel())
        with torch.cuda.device(self.device):
            if self.fit_hist is None:
                check(_lib.lib().gdpso_swarm_iterate(C.byref(self.desc), self.ev.gen.h, self.ev.ass.h, C.byref(run),
                                                     self._stream(), first, count))
            else:  # keep every iteration's fitness: one iteration per call, device-side copy in between
                for i in range(first, first + count):
                    check(_lib.lib().gdpso_swarm_iterate(C.byref(self.desc), self.ev.gen.h, self.ev.ass.h, C.byref(run),
                                                         self._stream(), i, 1))
                    self.fit_hist[i - 1].copy_(self.fitness)

    # -- read-back
    def control(self) -> np.ndarray:
        c = self.ctrl.cpu().numpy()
        check_chains(self.device)
        return c

    def results(self) -> dict:
        """Everything the reference's Swarm exposes after optimize(), as numpy arrays."""
        c = self.control()
        stop = int(c[_lib.CTRL_STOP_ITER])
        done = int(c[_lib.CTRL_ITERS_DONE])
        iters = stop if stop else done
        out = dict(iters=iters, stopped=bool(stop), w_inertia=float(c[_lib.CTRL_W]),
                   gbest_val=np.float32(c[_lib.CTRL_GBEST_VAL]), gbest_idx=int(c[_lib.CTRL_GBEST_IDX]),
                   g_best_val=self.g_best_val[:int(c[_lib.CTRL_N_IMPROVE])].cpu().numpy(),
                   g_best_val_dummy=self.g_best_val_dummy[:iters].cpu().numpy(),
                   g_best_pos=self.gbest_pos.cpu().numpy(), pos=self.pos.cpu().numpy(), vel=self.vel.cpu().numpy(),
                   pbest_pos=self.pbest_pos.cpu().numpy(), pbest_val=self.pbest_val.cpu().numpy())
        if self.mean_dist is not None:
            out["mean_mse"] = self.mean_dist[:iters].cpu().numpy()
        if self.hist_pos is not None:
            out["hist_pos"] = self.hist_pos[:iters + 1].cpu().numpy()
            out["hist_vel"] = self.hist_vel[:iters + 1].cpu().numpy()
        if self.fit_hist is not None:
            out["fitness"] = self.fit_hist[:iters].cpu().numpy()
        return out


class HipSwarmBatch:
    """Several independent swarms that share one evaluation batch (gdpso_swarm_iterate_multi): the
    reference runs one swarm per iid class, one after the other (src/training/pso_discovery.py:174-218);
    here an iteration of all of them is ONE generator + assessor pass over the concatenated
    particles.  Every swarm is a HipSwarm with its own state; positions and fitness values are
    slices of two shared buffers."""

    def __init__(self, evaluator: Evaluator, sizes, d, max_iters, *, history=True, diagnostics=True, record_fitness=False):
        """``history`` / ``diagnostics`` / ``record_fitness``: one flag for all the swarms or one per swarm."""
        self.ev, self.device, self.d = evaluator, evaluator.device, int(d)
        self.sizes = [int(n) for n in sizes]
        self.total = sum(self.sizes)
        self.pos_all = torch.zeros(self.total, d, dtype=torch.float64, device=self.device)
        self.fit_all = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.swarms, off = [], 0
        per_swarm = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * len(self.sizes)
        for n, t, h, dg, rf in zip(self.sizes, max_iters, per_swarm(history), per_swarm(diagnostics), per_swarm(record_fitness)):
            self.swarms.append(HipSwarm(evaluator, n, d, t, history=bool(h), diagnostics=bool(dg), record_fitness=bool(rf),
                                        pos=self.pos_all[off:off + n], fitness=self.fit_all[off:off + n]))
            off += n
        self.max_iter = max(int(t) for t in max_iters)
        self.class_per_image, self._class_key = None, None

    def iterate(self, first, count, specs, params):
        """Iterations first .. first+count-1 of every swarm.  specs: one FitnessSpec per swarm (mode,
        threshold and eps must agree; class_index may differ); params: per swarm a dict with w_cogn,
        w_soci, schedule_inertia, early_stopping, tolerance, diagnostics, seed."""
        img, img01, mse = self.ev.image_buffers(self.total)
        ws = self.ev.workspace(self.total)
        class_key = tuple(sp.class_index for sp in specs)
        if self._class_key != class_key:  # (a later call may come with other classes of interest)
            cls = np.concatenate([np.full(n, sp.class_index, np.int32) for n, sp in zip(self.sizes, specs)])
            self.class_per_image, self._class_key = torch.from_numpy(cls).to(self.device), class_key
        k = len(self.swarms)
        descs = (_lib.SwarmDesc * k)(*[s.desc for s in self.swarms])
        runs = (_lib.RunDesc * k)()
        for j, (s, sp, pr) in enumerate(zip(self.swarms, specs, params)):
            runs[j] = _lib.RunDesc(class_index=sp.class_index, mode=sp.mode_code(), threshold=sp.threshold, eps=sp.eps, target=0,
                                   w_ass=sp.w_ass, w_rec=sp.w_rec, w_cogn=pr["w_cogn"], w_soci=pr["w_soci"],
                                   schedule_inertia=int(pr["schedule_inertia"]), early_stopping=int(pr["early_stopping"]),
                                   tolerance=pr["tolerance"], rand=ptr(s.rand), seed=int(pr.get("seed", 0)),
                                   diagnostics=int(bool(pr.get("diagnostics", True)) and s.mean_dist is not None), f32_first_move=0,
                                   img01=ptr(img01), img=ptr(img), mse=ptr(mse), workspace=ptr(ws), workspace_bytes=ws.numel())
        record = any(s.fit_hist is not None for s in self.swarms)
        with torch.cuda.device(self.device):
            stream = _lib.stream_handle(self.device)
            step = 1 if record else count
            for i in range(first, first + count, step):
                check(_lib.lib().gdpso_swarm_iterate_multi(descs, runs, k, self.ev.gen.h, self.ev.ass.h, ptr(self.class_per_image),
                                                           stream, i, step))
                if record:
                    for s in self.swarms:
                        if s.fit_hist is not None and i <= s.max_iter:
                            s.fit_hist[i - 1].copy_(s.fitness)

    def run(self, specs, params, poll_every=8):
        """All iterations; with early stopping the host looks at the stop flags every few iterations
        and ends the loop once every swarm has stopped."""
        i = 1
        any_es = any(p["early_stopping"] for p in params)
        while i <= self.max_iter:
            chunk = min(poll_every, self.max_iter - i + 1) if any_es else self.max_iter - i + 1
            self.iterate(i, chunk, specs, params)
            i += chunk
            if any_es and i <= self.max_iter:
                done = [int(s.control()[_lib.CTRL_STOP_ITER]) != 0 or i > s.max_iter for s in self.swarms]
                if all(done):
                    break


# -------------------------------------------------------------------------- the loops
def shard_bounds(n_global: int, n_ranks: int, rank: int):
    """Contiguous particle blocks: rank r owns [lo, hi); the first n % W ranks get one more."""
    base, extra = divmod(n_global, n_ranks)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def run_loop(swarm, n_iterations, spec, w_cogn, w_soci, *, schedule_inertia=False, early_stopping=True, tolerance=1e-4,
             diagnostics=True, seed=0, comm=None, poll_every=8):
    """The while-loop of reference src/pso/util_pso.py:174-189.

    ``swarm`` is a HipSwarm (or, in the CPU tests of the sharded protocol, an object with
    the same phase methods).  With ``comm`` (a torch.distributed process group wrapper with
    ``all_gather(out, inp)``) particles are sharded over ranks and the only exchange per
    iteration is the all-gather of one {value, index, position} record per rank, which the
    optional diagnostic's positions ride along with (sharded_iterations).  Nothing here synchronises with the
    device except the periodic early-stop poll.
    """
    if comm is None:
        i = 1
        while i <= n_iterations:
            chunk = min(poll_every, n_iterations - i + 1) if early_stopping else n_iterations - i + 1
            swarm.iterate(i, chunk, spec, w_cogn, w_soci, schedule_inertia, early_stopping, tolerance, diagnostics, seed)
            i += chunk
            if early_stopping and i <= n_iterations and int(swarm.control()[_lib.CTRL_STOP_ITER]):
                break
        return
    sharded_iterations(swarm, comm, 1, n_iterations, spec, w_cogn, w_soci, schedule_inertia, early_stopping, tolerance,
                       diagnostics, seed, poll_every)


def sharded_iterations(swarm, comm, first, count, spec, w_cogn, w_soci, schedule_inertia, early_stopping, tolerance,
                       diagnostics, seed=0, poll_every=8, after_phase=None, check_every=64):
    """Iterations first .. first+count-1 of the sharded loop.  ONE exchange per iteration lies between
    a rank's fitness values and its move: the all-gather of the exchange table (the candidate records).

    The diagnostic of iteration i (mean pairwise distance of the positions after its move) needs every
    rank's positions; every rank then sums the distances of its share of the pairs into its candidate
    record, the sums travel with a later exchange, and every rank adds them in rank order.  Two forms:

    * ``swarm.table_positions`` (ranks of equal size): the positions lie behind the records in the
      table, so the exchange of iteration i+1 brings them, the sums follow with the exchange of
      iteration i+2, and there is no other collective; two more exchanges at the end flush the last
      two diagnostics.
    * otherwise the positions' all-gather starts right behind the move (large ones beside the next
      iteration's generator and assessor passes, TorchComm.start_gather_positions), the sums travel
      with the records of iteration i+1, and one more exchange at the end flushes the last diagnostic.

    ``after_phase(i, name)`` is a debugging hook.  Every ``poll_every`` iterations (early stopping) or
    ``check_every`` iterations (otherwise; 0: never) the host waits for the device and raises if a
    chained launch has given up since the last look, instead of computing on for the rest of the run.
    """
    hook = after_phase or (lambda i, name: None)
    table, row = swarm.exchange_table, swarm.exchange_table[swarm.rank]
    on_gpu = getattr(getattr(swarm, "device", None), "type", None) == "cuda"

    def stop_or_check(i):
        if early_stopping and i % poll_every == 0:
            return bool(int(swarm.control()[_lib.CTRL_STOP_ITER]))  # (control() looks at the chains too)
        if on_gpu and check_every and i % check_every == 0:
            check_chains(swarm.device)
        return False

    if diagnostics and getattr(swarm, "table_positions", False):
        moved = summed = 0  # last iteration whose move is done / whose sums wait in the records for an exchange
        for i in range(first, first + count):
            swarm.evaluate(i, spec)
            hook(i, "evaluate")
            swarm.local_best(i)
            hook(i, "local_best")
            comm.all_gather(table, row)  # records of i, positions after move(moved), sums of `summed`
            hook(i, "exchange")
            if moved:
                swarm.distance_part_table(moved)  # before this rank's own block moves on
            swarm.after_exchange(i, schedule_inertia, early_stopping, tolerance, w_cogn, w_soci, seed, summed)
            hook(i, "move")
            summed, moved = moved, i
            if stop_or_check(i):
                break
        if moved:
            comm.all_gather(table, row)
            if summed:
                swarm.distance_finish(summed)
            swarm.distance_part_table(moved)
            comm.all_gather(table, row)
            swarm.distance_finish(moved)
            hook(moved, "distance")
        return
    pending = None  # (iteration, handle of the position all-gather started behind its move)
    for i in range(first, first + count):
        swarm.evaluate(i, spec)
        hook(i, "evaluate")
        swarm.local_best(i)
        hook(i, "local_best")
        if pending is not None:
            swarm.distance_part(pending[0], pending[1].wait())
        comm.all_gather(table, row)
        hook(i, "exchange")
        swarm.after_exchange(i, schedule_inertia, early_stopping, tolerance, w_cogn, w_soci, seed, pending[0] if pending is not None else 0)
        pending = None
        hook(i, "move")
        if diagnostics:
            pending = (i, comm.start_gather_positions(swarm.pos, swarm.n_global))
        if stop_or_check(i):
            break
    if pending is not None:
        swarm.distance_part(pending[0], pending[1].wait())
        comm.all_gather(table, row)
        swarm.distance_finish(pending[0])
        hook(pending[0], "distance")


class _Gathered:
    """Handle of a position all-gather in flight; wait() returns the positions in global particle
    order (a buffer of the TorchComm that its next position gather overwrites)."""

    def __init__(self, work, result):
        self._work, self._result = work, result

    def wait(self) -> torch.Tensor:
        if self._work is not None:  # (None: the collective ran in stream order, or on the host)
            self._work.wait()  # device tensors: the CURRENT STREAM waits, not the host
            self._work = None
        return self._result() if callable(self._result) else self._result


class TorchComm:
    """torch.distributed plumbing for the sharded loop ("nccl" is RCCL on ROCm; "gloo" in CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # gloo has no stream-ordered collectives on device tensors: with it (CPU tests, rehearsal of
        # several ranks on one GPU) device tensors are staged through the host explicitly.  RCCL
        # collectives run in stream order on the device tensors themselves.
        self.host_staged = dist.get_backend(group) == "gloo"
        self._row = None
        # Position all-gathers of at least this many bytes (all ranks together) go to the backend's own
        # stream and run beside the next iteration (two stream hand-shakes: ~25 us of stream time
        # whatever the size); smaller ones stay on the launch stream like the record exchange
        # (~8 us plus the transfer itself).
        self.async_bytes = int(os.environ.get("GDPSO_COMM_ASYNC_BYTES", str(1 << 20)))

    def _all_gather_flat(self, out_flat: torch.Tensor, inp_flat: torch.Tensor, async_op=False):
        if self.host_staged and out_flat.is_cuda:
            host_in = inp_flat.detach().cpu()  # waits for the producing kernels
            host_out = torch.empty(out_flat.numel(), dtype=out_flat.dtype)
            self.dist.all_gather_into_tensor(host_out, host_in, group=self.group)
            out_flat.copy_(host_out)
            return None
        return self.dist.all_gather_into_tensor(out_flat, inp_flat, group=self.group, async_op=async_op)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        """out [W, ...]: the table of candidate records; inp: this rank's row OF THAT TABLE.  Over RCCL
        the collective runs in place (its input is the rank's slot of its output, the form the library
        documents for all-gather), in stream order on the current stream; gloo gets a copy of the row."""
        if self.host_staged:
            if self._row is None or self._row.shape != inp.shape or self._row.device != inp.device:
                self._row = torch.empty_like(inp)
            self._row.copy_(inp)
            inp = self._row
        self._all_gather_flat(out.view(-1), inp.reshape(-1))

    def _pos_buffers(self, pos: torch.Tensor, n_global: int):
        n_max = -(-n_global // self.world)
        d = pos.shape[1]
        key = (n_max, d, pos.dtype, pos.device)
        if getattr(self, "_pos_key", None) != key:
            self._pos_key = key
            self._pos_all = torch.empty(self.world * n_max, d, dtype=pos.dtype, device=pos.device)
            self._pos_pad = torch.zeros(n_max, d, dtype=pos.dtype, device=pos.device)
        return n_max, self._pos_all

    def start_gather_positions(self, pos: torch.Tensor, n_global: int) -> _Gathered:
        """Begin the all-gather of every rank's positions and return at once.  Over RCCL a large one
        (async_bytes) runs on the backend's own stream behind everything enqueued so far on the current
        one, so the kernels enqueued next run beside it; ``pos`` must stay as it is until wait() has been
        called on the handle (the loop's next move comes later than that).  Unequal shards go out as padded
        blocks and are put in global particle order by wait()."""
        n_max, allp = self._pos_buffers(pos, n_global)
        beside = allp.numel() * allp.element_size() >= self.async_bytes
        if n_global == self.world * n_max:  # equal shards: gather straight from the swarm state
            return _Gathered(self._all_gather_flat(allp.view(-1), pos.reshape(-1), async_op=beside), allp)
        self._pos_pad[:pos.shape[0]] = pos
        work = self._all_gather_flat(allp.view(-1), self._pos_pad.view(-1), async_op=beside)

        def assemble():
            rows = []
            for r in range(self.world):
                lo, hi = shard_bounds(n_global, self.world, r)
                rows.append(allp[r * n_max:r * n_max + (hi - lo)])
            return torch.cat(rows).contiguous()

        return _Gathered(work, assemble)

    def gather_positions(self, pos: torch.Tensor, n_global: int) -> torch.Tensor:
        """Every rank's positions, in global particle order.  The returned tensor is a buffer of
        this object that the next position gather overwrites."""
        return self.start_gather_positions(pos, n_global).wait()

    def gather_rows(self, local: np.ndarray, axis: int, n_global: int) -> np.ndarray:
        """Concatenate per-rank numpy blocks along ``axis`` (end-of-run history collection)."""
        objs = [None] * self.world
        self.dist.all_gather_object(objs, local, group=self.group)
        return np.concatenate(objs, axis=axis)
