"""The allocation-free nearest-code search (gdpso_vq_nearest_static) against gdpso_vq_nearest,
through the C ABI: two calls in a row, the second with a smaller input and a smaller codebook,
bit for bit (indices, gathered codes, best distances)."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, x, cb):
    L = lib.lib()
    n, d = x.shape
    k = cb.shape[0]
    dev = x.device
    outs = []
    for static in (False, True):
        idx = torch.full((n,), -7, dtype=torch.int64, device=dev)
        codes = torch.full((n, d), float("nan"), dtype=torch.float32, device=dev)
        dist = torch.full((n,), float("nan"), dtype=torch.float32, device=dev)
        stream = lib.stream_handle(dev)
        if static:
            lib.check(L.gdpso_vq_nearest_static(stream, n, d, k, x.data_ptr(), cb.data_ptr(), idx.data_ptr(), codes.data_ptr(),
                                                dist.data_ptr()))
        else:
            ws = torch.empty(L.gdpso_vq_workspace_bytes(k, d), dtype=torch.uint8, device=dev)
            lib.check(L.gdpso_vq_nearest(stream, n, d, k, x.data_ptr(), cb.data_ptr(), idx.data_ptr(), codes.data_ptr(),
                                         dist.data_ptr(), ws.data_ptr(), ws.numel()))
        torch.cuda.synchronize(dev)
        outs.append((idx.cpu().numpy(), codes.cpu().numpy(), dist.cpu().numpy()))
    return outs


def test_static_scratch_search_matches_workspace_search():
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    # first call: the reference workflow's codebook (256 codes of 100), 1,001 rows (a ragged last workgroup)
    cb1 = torch.randn(256, 100, generator=g).to(dev)
    x1 = torch.randn(1001, 100, generator=g).to(dev)
    x1[:16] = cb1[40:56]  # exact hits
    # second call: smaller input, smaller codebook with an odd code count and a duplicated row
    cb2 = torch.randn(37, 24, generator=g)
    cb2[30] = cb2[3]
    cb2 = cb2.to(dev)
    x2 = torch.randn(10, 24, generator=g).to(dev)
    x2[4] = cb2[3]
    for x, cb in ((x1, cb1), (x2, cb2)):
        (i0, c0, d0), (i1, c1, d1) = _both(lib, x, cb)
        assert i0.min() >= 0 and i0.max() < cb.shape[0]
        np.testing.assert_array_equal(i1, i0)
        np.testing.assert_array_equal(c1, c0)
        np.testing.assert_array_equal(d1, d0)
    assert int(_both(lib, x2, cb2)[1][0][4]) == 3  # the lower of two identical codes
    # beyond the static limit: refused with a clear error, nothing launched
    L = lib.lib()
    assert L.gdpso_vq_nearest_static(None, 1, 100, 4096, x1.data_ptr(), cb1.data_ptr(), x1.data_ptr(), None, None) == 3
    assert b"static scratch" in L.gdpso_last_error()
