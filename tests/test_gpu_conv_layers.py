"""One convolution layer at a time through the C ABI (gdpso_conv_layer_run), with the tile
shape and the split count forced, against torch's float64 convolution of the same data.

Whole-network tests reach a tile shape only when the planner picks it (the 128x128 tile, for
one, needs more work than any test-sized network offers); here every instantiation of the
kernel runs on shapes that exercise row tails, partly filled phases, border classes of padded
windows, strides, the four sub-pixel phases of the transposed convolution, residuals and every
activation.  Tolerance: 1e-4 absolute on outputs of unit scale (fp32 accumulation over K up
to 4608 terms stays below 2e-5 here)."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-4
T128, T64, T32, T16 = 128, 64, 32, 16


@pytest.fixture(scope="module")
def lib():
    return importlib.import_module("gan-discovery-pso_amd._lib")


TORCH_DTYPE = {"bf16": torch.bfloat16, "f16": torch.float16}
# relative bound of the final rounding to the storage type (half an ulp: 2^-9 bf16, 2^-11 fp16), doubled
RTOL16 = {"bf16": 2.0 ** -8, "f16": 2.0 ** -10}


def run_layer(lib, n, hin, cin, cout, k, stride, pad, transposed, act, tile, splits, with_res, seed, dtype="f32"):
    """dtype "bf16" / "f16": the layer runs on inputs, weights and residual rounded to that type (the
    reference below uses the same rounded values in float64, so only the fp32 accumulation and the
    final rounding of the output separate the two)."""
    rng = np.random.default_rng(seed)
    rnd = (lambda a: a) if dtype == "f32" else (lambda a: torch.from_numpy(a).to(TORCH_DTYPE[dtype]).float().numpy())
    taps = 16 if transposed else k * k
    wshape = (cin, cout, 4, 4) if transposed else (cout, cin, k, k)
    w = (rng.standard_normal(wshape) / np.sqrt(cin * taps / (4 if transposed else 1))).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
    x = rng.standard_normal((n, cin, hin, hin)).astype(np.float32)
    xt = torch.from_numpy(rnd(x)).double()
    if transposed:
        y = F.conv_transpose2d(xt, torch.from_numpy(rnd(w)).double(), stride=2, padding=1)
    else:
        y = F.conv2d(xt, torch.from_numpy(rnd(w)).double(), stride=stride, padding=pad)
    y = y * torch.from_numpy(scale).double()[None, :, None, None] + torch.from_numpy(shift).double()[None, :, None, None]
    res = None
    if with_res:
        res = rng.standard_normal(tuple(y.shape)).astype(np.float32)
        y = y + torch.from_numpy(rnd(res)).double()
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = F.leaky_relu(y, 0.2)
    want = y.permute(0, 2, 3, 1).contiguous().numpy()

    d_in = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cuda()
    d_res = torch.from_numpy(np.ascontiguousarray(res.transpose(0, 2, 3, 1))).cuda() if with_res else None
    d_out = torch.full(want.shape, float("nan"), dtype=torch.float32, device="cuda")
    tile_used, splits_used = C.c_int(0), C.c_int(0)
    torch.cuda.synchronize()
    args = (n, hin, cin, cout, k, stride, pad, int(transposed), act, tile, splits,
            w.ctypes.data, scale.ctypes.data, shift.ctypes.data,
            d_in.data_ptr(), lib.ptr(d_res), d_out.data_ptr(), C.byref(tile_used), C.byref(splits_used))
    if dtype == "f32":
        lib.check(lib.lib().gdpso_conv_layer_run(*args))
    else:
        lib.check(lib.lib().gdpso_conv_layer_run_typed(*args, lib.DTYPES[dtype]))
    got = d_out.cpu().numpy()
    assert np.isfinite(got).all(), "some output element was never written"
    if dtype == "f32":
        np.testing.assert_allclose(got, want, atol=TOL, rtol=0)
    else:
        np.testing.assert_allclose(got, want, atol=TOL, rtol=RTOL16[dtype])
    return tile_used.value, splits_used.value


# (n, hin, cin, cout, k, stride, pad): ResNet-like layers at sizes with row tails
SHAPES = [
    (3, 7, 128, 128, 3, 1, 1),    # 3x3 pad 1: nine border classes, 147 rows (tail in every tile)
    (5, 14, 256, 128, 1, 1, 0),   # 1x1
    (2, 14, 128, 256, 1, 2, 0),   # strided 1x1 (identity_downsample)
    (2, 14, 128, 128, 3, 2, 1),   # strided 3x3: border classes on the first row/column only
    (1, 4, 256, 128, 3, 1, 1),    # tiny map: the middle class is 2x2
    (1, 1, 128, 128, 3, 1, 1),    # 1x1 map with padding: a single class with one live tap
    (7, 5, 128, 128, 3, 1, 0),    # unpadded window
]


@pytest.mark.parametrize("tile", [T128, T64, T32, T16])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_every_tile_shape(lib, tile, shape):
    n, hin, cin, cout, k, stride, pad = shape
    t, s = run_layer(lib, n, hin, cin, cout, k, stride, pad, False, act=1, tile=tile, splits=1, with_res=True,
                     seed=hash((tile,) + shape) & 0xffff)
    assert (t, s) == (tile, 1)


@pytest.mark.parametrize("tile", [T128, T64, T32, T16])
@pytest.mark.parametrize("splits", [2, 3, 5, 16])
def test_split_k(lib, tile, splits):
    """K split over blockIdx.z into slabs (uneven last part; 16 leaves some parts empty at K=128)."""
    for shape, act, res in (((2, 7, 128, 128, 3, 1, 1), 1, True), ((3, 5, 128, 256, 1, 1, 0), 0, False)):
        t, s = run_layer(lib, *shape, False, act=act, tile=tile, splits=splits, with_res=res, seed=tile * 31 + splits)
        assert (t, s) == (tile, splits)


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("with_res", [False, True])
def test_activations_and_residual(lib, act, with_res):
    run_layer(lib, 4, 7, 64, 64, 3, 1, 1, False, act=act, tile=0, splits=0, with_res=with_res, seed=act * 2 + with_res)


@pytest.mark.parametrize("tile", [0, T128, T64, T32, T16])
def test_transposed_k4s2p1(lib, tile):
    """ConvTranspose2d(k4, s2, p1): four sub-pixel phases, each with its own border classes."""
    run_layer(lib, 3, 7, 128, 128, 4, 2, 1, True, act=2, tile=tile, splits=1, with_res=False, seed=77 + tile)
    run_layer(lib, 2, 3, 128, 64 if tile in (0, T64, T32, T16) else 128, 4, 2, 1, True, act=0, tile=tile, splits=2,
              with_res=False, seed=78 + tile)


def test_planner_choice_on_wide_layers(lib):
    """Layers big enough for the planner itself to pick the 128x128 tile, and deep K."""
    t, _ = run_layer(lib, 192, 14, 256, 512, 1, 1, 0, False, act=1, tile=0, splits=0, with_res=True, seed=5)
    assert t == T128
    run_layer(lib, 2, 4, 512, 64, 3, 1, 1, False, act=1, tile=0, splits=0, with_res=False, seed=6)  # K = 4608


def test_rejects_bad_arguments(lib):
    x = torch.zeros(1, 4, 4, 32, device="cuda")
    out = torch.zeros(1, 4, 4, 64, device="cuda")
    w = np.zeros((64, 32, 1, 1), np.float32)
    v = np.zeros(64, np.float32)
    call = lib.lib().gdpso_conv_layer_run
    # tile 32 needs cin % 64 == 0
    assert call(1, 4, 32, 64, 1, 1, 0, 0, 0, T32, 1, w.ctypes.data, v.ctypes.data, v.ctypes.data, x.data_ptr(), 0,
                out.data_ptr(), None, None) != 0
    # cout must be a multiple of 64
    assert call(1, 4, 32, 48, 1, 1, 0, 0, 0, 0, 0, w.ctypes.data, v.ctypes.data, v.ctypes.data, x.data_ptr(), 0,
                out.data_ptr(), None, None) != 0
    # null input
    assert call(1, 4, 32, 64, 1, 1, 0, 0, 0, 0, 0, w.ctypes.data, v.ctypes.data, v.ctypes.data, 0, 0,
                out.data_ptr(), None, None) != 0


def test_random_layer_shapes(lib):
    """Seeded sweep over window sizes, strides, paddings, map sizes, channel counts, tile shapes
    and split counts (the encoder's k4 s2 p1 and the 5x5 / 7x7 windows among them)."""
    rng = np.random.default_rng(2024)
    done = 0
    while done < 48:
        k = int(rng.choice([1, 2, 3, 4, 5, 7]))
        stride = int(rng.choice([1, 2, 3]))
        pad = int(rng.integers(0, k // 2 + 1))
        hin = int(rng.integers(max(1, k - 2 * pad), 12))
        cin = int(rng.choice([32, 64, 128, 160, 256]))
        cout = int(rng.choice([64, 128, 192]))
        n = int(rng.integers(1, 6))
        tiles = [T64] + ([T32] if cin % 64 == 0 else []) + ([T16] if cin % 128 == 0 else []) + ([T128] if cout % 128 == 0 else [])
        tile = int(rng.choice(tiles))
        splits = int(rng.choice([1, 1, 2, 3]))
        act = int(rng.integers(0, 3))
        with_res = bool(rng.integers(0, 2))
        try:
            run_layer(lib, n, hin, cin, cout, k, stride, pad, False, act=act, tile=tile, splits=splits, with_res=with_res,
                      seed=1000 + done)
        except AssertionError as e:
            raise AssertionError(f"case n={n} hin={hin} cin={cin} cout={cout} k={k} stride={stride} pad={pad} tile={tile} "
                                 f"splits={splits} act={act} res={with_res}: {e}") from e
        done += 1


# ----------------------------------------------------------------- 16-bit storage (bf16 / fp16 matrix instructions)
# K-tiles are counted in 16-byte vectors, so the channel quanta double: cin % 64 for the 64x64 and
# 128x128 tiles, % 128 for the 32x64 tile, % 256 for the 32x32 tile.
SHAPES16 = [
    (3, 7, 256, 128, 3, 1, 1),    # 3x3 pad 1: nine border classes, row tails
    (5, 14, 256, 128, 1, 1, 0),   # 1x1
    (2, 14, 256, 256, 1, 2, 0),   # strided 1x1
    (2, 14, 256, 128, 3, 2, 1),   # strided 3x3
    (1, 1, 256, 128, 3, 1, 1),    # 1x1 map with padding: one live tap
    (7, 5, 512, 128, 3, 1, 0),    # unpadded window
]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("tile", [T128, T64, T32, T16])
@pytest.mark.parametrize("shape", SHAPES16, ids=lambda s: "x".join(map(str, s)))
def test_every_tile_shape_16bit(lib, dtype, tile, shape):
    n, hin, cin, cout, k, stride, pad = shape
    t, s = run_layer(lib, n, hin, cin, cout, k, stride, pad, False, act=1, tile=tile, splits=1, with_res=True,
                     seed=hash((tile,) + shape) & 0xffff, dtype=dtype)
    assert (t, s) == (tile, 1)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("tile", [T128, T64, T32, T16])
@pytest.mark.parametrize("splits", [2, 3, 16])
def test_split_k_16bit(lib, dtype, tile, splits):
    for shape, act, res in (((2, 7, 256, 128, 3, 1, 1), 1, True), ((3, 5, 256, 256, 1, 1, 0), 0, False)):
        t, s = run_layer(lib, *shape, False, act=act, tile=tile, splits=splits, with_res=res, seed=tile * 31 + splits, dtype=dtype)
        assert (t, s) == (tile, splits)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("tile", [0, T128, T64, T32, T16])
def test_transposed_k4s2p1_16bit(lib, dtype, tile):
    run_layer(lib, 3, 7, 256, 128, 4, 2, 1, True, act=2, tile=tile, splits=1, with_res=False, seed=77 + tile, dtype=dtype)
    run_layer(lib, 2, 3, 128, 64, 4, 2, 1, True, act=1, tile=0, splits=0, with_res=False, seed=79 + tile, dtype=dtype)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_random_layer_shapes_16bit(lib, dtype):
    rng = np.random.default_rng(4048)
    for done in range(32):
        k = int(rng.choice([1, 2, 3, 4, 5, 7]))
        stride = int(rng.choice([1, 2, 3]))
        pad = int(rng.integers(0, k // 2 + 1))
        hin = int(rng.integers(max(1, k - 2 * pad), 12))
        cin = int(rng.choice([64, 128, 192, 256]))
        cout = int(rng.choice([64, 128, 192]))
        n = int(rng.integers(1, 6))
        tiles = [T64] + ([T32] if cin % 128 == 0 else []) + ([T16] if cin % 256 == 0 else []) + ([T128] if cout % 128 == 0 else [])
        tile = int(rng.choice(tiles))
        splits = int(rng.choice([1, 1, 2, 3]))
        act = int(rng.integers(0, 3))
        with_res = bool(rng.integers(0, 2))
        try:
            run_layer(lib, n, hin, cin, cout, k, stride, pad, False, act=act, tile=tile, splits=splits, with_res=with_res,
                      seed=3000 + done, dtype=dtype)
        except AssertionError as e:
            raise AssertionError(f"case {dtype} n={n} hin={hin} cin={cin} cout={cout} k={k} stride={stride} pad={pad} tile={tile} "
                                 f"splits={splits} act={act} res={with_res}: {e}") from e


def test_16bit_rejects_narrow_channels(lib):
    x = torch.zeros(1, 4, 4, 32, device="cuda")
    out = torch.zeros(1, 4, 4, 64, device="cuda")
    w = np.zeros((64, 32, 1, 1), np.float32)
    v = np.zeros(64, np.float32)
    assert lib.lib().gdpso_conv_layer_run_typed(1, 4, 32, 64, 1, 1, 0, 0, 0, 0, 0, w.ctypes.data, v.ctypes.data, v.ctypes.data,
                                                x.data_ptr(), 0, out.data_ptr(), None, None, lib.DTYPES["bf16"]) != 0
