"""ctypes binding of libgdpso.so (the C ABI declared in include/gdpso.h).

There is no fallback: if the HIP library is missing or fails to load, importing anything
that computes raises.  ``import torch`` happens first so that the HIP runtime already
mapped by PyTorch (soname libamdhip64.so.7) is the one the library binds to.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgdpso.so")

ABI_VERSION = 3
MODE_IN_TRAINING, MODE_OUT_TRAINING = 0, 1
DTYPES = {"f32": 0, "bf16": 1, "f16": 2}  # GDPSO_DTYPE_*
CTRL_DOUBLES = 8


def cand_doubles(d: int) -> int:
    """GDPSO_CAND_DOUBLES(d): doubles per candidate record {value, global index, position[d], distance sums[2]}."""
    return d + 4

CTRL_W, CTRL_GBEST_VAL, CTRL_GBEST_IDX, CTRL_N_IMPROVE, CTRL_STOP_ITER, CTRL_ITERS_DONE = range(6)

c_void_p, c_int, c_size_t, c_double, c_float, c_uint64 = C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_float, C.c_uint64
FloatPP = C.POINTER(C.POINTER(C.c_float))


class SwarmDesc(C.Structure):
    """gdpso_swarm of include/gdpso.h."""
    _fields_ = [
        ("n", c_int), ("d", c_int), ("first_index", c_int), ("n_ranks", c_int),
        ("pos", c_void_p), ("vel", c_void_p), ("pbest_pos", c_void_p), ("pbest_val", c_void_p),
        ("fitness", c_void_p), ("gbest_pos", c_void_p), ("ctrl", c_void_p),
        ("g_best_val", c_void_p), ("g_best_val_dummy", c_void_p), ("mean_dist", c_void_p),
        ("hist_pos", c_void_p), ("hist_vel", c_void_p), ("cand", c_void_p),
        ("scratch", c_void_p), ("scratch_bytes", c_size_t), ("max_iter", c_int), ("cand_stride", c_int),
    ]


class RunDesc(C.Structure):
    """gdpso_run of include/gdpso.h."""
    _fields_ = [
        ("class_index", c_int), ("mode", c_int), ("threshold", c_float), ("eps", c_float),
        ("target", c_void_p), ("w_ass", c_float), ("w_rec", c_float),
        ("w_cogn", c_double), ("w_soci", c_double),
        ("schedule_inertia", c_int), ("early_stopping", c_int), ("tolerance", c_double),
        ("rand", c_void_p), ("seed", c_uint64), ("diagnostics", c_int), ("f32_first_move", c_int),
        ("img01", c_void_p), ("img", c_void_p), ("mse", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
    ]


_PROTOTYPES = {
    "gdpso_abi_version": (c_int, []),
    "gdpso_last_error": (C.c_char_p, []),
    "gdpso_struct_size": (c_size_t, [c_int]),
    "gdpso_generator_create": (c_int, [C.POINTER(c_void_p), c_int, c_int, c_int, FloatPP, c_int]),
    "gdpso_generator_create_typed": (c_int, [C.POINTER(c_void_p), c_int, c_int, c_int, FloatPP, c_int, c_int]),
    "gdpso_generator_destroy": (None, [c_void_p]),
    "gdpso_generator_image_side": (c_int, [c_void_p]),
    "gdpso_generator_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "gdpso_generator_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_size_t]),
    "gdpso_assessor_create_resnet": (c_int, [C.POINTER(c_void_p), C.POINTER(c_int), c_int, c_int, FloatPP, c_int]),
    "gdpso_assessor_create_resnet_typed": (c_int, [C.POINTER(c_void_p), C.POINTER(c_int), c_int, c_int, FloatPP, c_int, c_int]),
    "gdpso_assessor_create_alexnet": (c_int, [C.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, FloatPP, c_int, c_int]),
    "gdpso_assessor_destroy": (None, [c_void_p]),
    "gdpso_assessor_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "gdpso_assessor_n_class": (c_int, [c_void_p]),
    "gdpso_assessor_flops_per_image": (c_double, [c_void_p]),
    "gdpso_assessor_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                       c_float, c_void_p, c_float, c_float, c_void_p, c_size_t]),
    "gdpso_encoder_create": (c_int, [C.POINTER(c_void_p), c_int, c_int, FloatPP, c_int]),
    "gdpso_encoder_destroy": (None, [c_void_p]),
    "gdpso_encoder_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "gdpso_encoder_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t]),
    "gdpso_vq_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gdpso_vq_seed_codebook": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int]),
    "gdpso_vq_nearest": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "gdpso_vq_lookup": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gdpso_vq_nearest_static": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gdpso_swarm_scratch_bytes": (c_size_t, [c_int]),
    "gdpso_swarm_reset": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_double]),
    "gdpso_swarm_local_best": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_int, c_int]),
    "gdpso_swarm_global_best": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_int, c_int, c_int, c_double]),
    "gdpso_swarm_move": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_int, c_double, c_double, c_void_p, c_uint64, c_int]),
    "gdpso_swarm_mean_distance": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_int, c_void_p, c_int]),
    "gdpso_swarm_distance_part": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_int, c_void_p, c_int, c_int]),
    "gdpso_swarm_distance_part_table": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_int, c_int, c_int]),
    "gdpso_swarm_distance_finish": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_int, c_int]),
    "gdpso_swarm_after_exchange": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_int, c_int, c_int, c_double, c_double, c_double, c_void_p,
                                           c_uint64, c_int, c_int, c_int]),
    "gdpso_conv_timing_begin": (c_int, [c_int]),
    "gdpso_conv_timing_end": (c_int, [C.POINTER(c_double), C.POINTER(c_int), C.POINTER(c_double)]),
    "gdpso_conv_timing_bytes": (c_double, []),
    "gdpso_conv_timing_executed_flops": (c_double, []),
    "gdpso_conv_plan_describe": (c_int, [c_int] * 9 + [C.POINTER(c_int), c_int]),
    "gdpso_fast_div_check": (C.c_uint, [C.c_uint, C.c_uint]),
    "gdpso_chain_debug": (c_int, [c_int, C.c_uint, c_int]),
    "gdpso_chain_stats": (c_int, [C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "gdpso_plan_options": (c_int, [c_int, c_int]),
    "gdpso_resident_options": (c_int, [c_int, c_int, c_int]),
    "gdpso_resident_force_images": (c_int, [c_int]),
    "gdpso_resident_generator_option": (c_int, [c_int]),
    "gdpso_resident_describe": (c_int, [c_int, c_int, c_int] + [C.POINTER(c_int)] * 4),
    "gdpso_chain_tile_describe": (c_int, [c_int] * 10),
    "gdpso_conv_probe": (c_int, [c_int] * 11 + [C.POINTER(c_double), C.POINTER(c_int), C.POINTER(c_int)]),
    "gdpso_conv_probe_typed": (c_int, [c_int] * 11 + [C.POINTER(c_double), C.POINTER(c_int), C.POINTER(c_int), c_int]),
    "gdpso_conv_layer_run": (c_int, [c_int] * 11 + [c_void_p] * 6 + [C.POINTER(c_int), C.POINTER(c_int)]),
    "gdpso_conv_layer_run_typed": (c_int, [c_int] * 11 + [c_void_p] * 6 + [C.POINTER(c_int), C.POINTER(c_int), c_int]),
    "gdpso_assessor_forward_classes": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                               c_float, c_void_p, c_float, c_float, c_void_p, c_size_t]),
    "gdpso_swarm_iterate_multi": (c_int, [C.POINTER(SwarmDesc), C.POINTER(RunDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int]),
    "gdpso_swarm_iterate": (c_int, [C.POINTER(SwarmDesc), c_void_p, c_void_p, C.POINTER(RunDesc), c_void_p, c_int, c_int]),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


class GdpsoError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GdpsoError(f"{LIB_PATH} not found: build it with `python gan-discovery-pso_amd/build.py` "
                             "(hipcc, gfx950); this package has no CPU fallback")
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.gdpso_abi_version() != ABI_VERSION:
            raise GdpsoError("libgdpso.so ABI version mismatch: rebuild the library")
        if handle.gdpso_struct_size(0) != C.sizeof(SwarmDesc) or handle.gdpso_struct_size(1) != C.sizeof(RunDesc):
            raise GdpsoError("struct layout mismatch between _lib.py and libgdpso.so")
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().gdpso_last_error()
        raise GdpsoError(f"libgdpso error {rc}: {msg.decode() if msg else '?'}")


def host_float_pointers(arrays):
    """List of contiguous float32 numpy arrays -> (float** for the C ABI, keep-alive list)."""
    import numpy as np
    keep = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
    arr = (C.POINTER(C.c_float) * len(keep))(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in keep])
    return arr, keep


def stream_handle(device=None) -> int:
    """Raw hipStream_t of torch's current stream on ``device``."""
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
