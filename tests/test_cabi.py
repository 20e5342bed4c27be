"""The C ABI without a GPU: the library loads, exports every symbol the header declares,
and rejects bad arguments before touching the device."""
import ctypes as C
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return importlib.import_module("gan-discovery-pso_amd._lib")


def header_functions():
    text = open(os.path.join(ROOT, "include", "gdpso.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gdpso_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    declared = header_functions()
    assert len(declared) >= 25
    assert sorted(lib.EXPORTED_SYMBOLS) == declared


def test_every_declared_symbol_is_exported(lib):
    handle = lib.lib()
    for name in header_functions():
        assert getattr(handle, name) is not None
    assert handle.gdpso_abi_version() == lib.ABI_VERSION == 3


def test_struct_layouts_match_header(lib):
    h = lib.lib()
    assert h.gdpso_struct_size(0) == C.sizeof(lib.SwarmDesc)
    assert h.gdpso_struct_size(1) == C.sizeof(lib.RunDesc)
    assert h.gdpso_struct_size(2) == 0


def test_argument_errors_without_gpu(lib):
    h = lib.lib()
    assert h.gdpso_swarm_scratch_bytes(0) == 0
    assert h.gdpso_swarm_scratch_bytes(512) >= 512 // 256 * 8
    assert h.gdpso_generator_workspace_bytes(None, 8) == 0
    assert h.gdpso_assessor_workspace_bytes(None, 8) == 0
    assert h.gdpso_generator_image_side(None) == 0
    assert h.gdpso_generator_forward(None, None, 1, None, 1, None, None, None, None, None, 0) == 1
    assert b"null argument" in h.gdpso_last_error()
    assert h.gdpso_assessor_forward(None, None, 1, None, None, None, 0, 1, 0.0, 0.1, None, 1.0, 1.0, None, 0) == 1
    desc = lib.SwarmDesc(n=0, d=4, n_ranks=1, max_iter=1)
    assert h.gdpso_swarm_reset(C.byref(desc), None, 0.5) == 1
    assert b"must be positive" in h.gdpso_last_error()
    desc = lib.SwarmDesc(n=4, d=4, n_ranks=1, max_iter=1)
    assert h.gdpso_swarm_move(C.byref(desc), None, 1, 1.0, 1.0, None, 0, 0) == 1
    assert b"null state buffer" in h.gdpso_last_error()
    out = C.c_void_p()
    assert h.gdpso_generator_create(C.byref(out), 10, 64, 7, None, 14) == 1
    depths = (C.c_int * 4)(3, 4, 6, 3)
    arr = (C.POINTER(C.c_float) * 1)()
    assert h.gdpso_assessor_create_resnet(C.byref(out), depths, 8, 28, arr, 1) == 1
    assert b"wrong number of tensors" in h.gdpso_last_error()
    # the allocation-free search refuses sizes beyond its static scratch before touching the device
    assert h.gdpso_vq_nearest_static(None, 1, 100, 4096, 1, 1, 1, None, None) == 3
    assert b"static scratch" in h.gdpso_last_error()
    with pytest.raises(lib.GdpsoError):
        lib.check(1)
