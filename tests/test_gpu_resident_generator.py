"""The generator of the reference's shape as one launch (a workgroup keeps one or two latents' 7x7x128 and
14x14x64 maps in LDS) against the tiled kernels of the same handle, the reference's own activations
(tests/golden/layers.npz) and the oracle.  First and last layer add in the tiled kernels' order; the middle
layer's sums are ordered differently, so the two paths agree to fp32 rounding, not bit for bit."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4
# what the process started with (tools/switch_matrix.sh runs the suite under the environment switches): restored after every case
ENV_GEN = int(os.environ.get("GDPSO_GEN_RESIDENT", "1")[:1])
ENV_STAGES = 0 if os.environ.get("GDPSO_STAGE1_RESIDENT", "1") == "0" else 1 if os.environ.get("GDPSO_STAGE2_RESIDENT", "1") == "0" else 2
ENV_MAX = int(os.environ.get("GDPSO_STAGE1_RESIDENT_MAX", "-1"))  # (-1: the library's default)
ENV_MIN = int(os.environ.get("GDPSO_STAGE1_RESIDENT_MIN", "-1"))


@pytest.fixture(scope="module")
def eng():
    return importlib.import_module("gan-discovery-pso_amd.engine")


@pytest.fixture(scope="module")
def lib():
    return importlib.import_module("gan-discovery-pso_amd._lib")


def _evaluator(eng, synth, zd, dtype="f32"):
    dev = torch.device("cuda:0")
    gen = eng.PackedGenerator(synth.generator_state(zd, 1, 64, seed=11), dev, dtype=dtype)
    ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=28, dtype=dtype)
    return eng.Evaluator(gen, ass)


def _images(eng, ev, z, resident, target=None, min_images=1):
    """(img, img01, mse | None) as numpy, the resident launch on or off at every batch size from min_images on."""
    eng.resident_options(2, 1 << 30, min_images)
    # (True: at every batch size within the limits, not only where the launch pays; an int: that mode of the option)
    eng.resident_generator((2 if resident else 0) if isinstance(resident, bool) else resident)
    try:
        out = []
        for _ in range(2):
            img01, img, mse = ev.images(z, target)
            out.append((img.cpu().numpy().copy(), img01.cpu().numpy().copy(), None if mse is None else mse.cpu().numpy().copy()))
    finally:
        eng.resident_generator(ENV_GEN)
        eng.resident_options(ENV_STAGES, ENV_MAX, ENV_MIN)
    for a, b in zip(out[0], out[1]):
        if a is not None:
            np.testing.assert_array_equal(a, b)
    return out[0]


@pytest.mark.parametrize("tag,zd", [("z10", 10), ("z100", 100)])
def test_reference_activations(eng, synth, tag, zd):
    """The reference's own latents and images: short latents (first layer inside the launch) and long ones (first
    layer as a GEMM launch in front)."""
    g = load_golden("layers")
    ev = _evaluator(eng, synth, zd)
    z = torch.from_numpy(g[tag + "_z"]).cuda()
    img, img01, _ = _images(eng, ev, z, True)
    np.testing.assert_allclose(img, g[tag + "_img"][:, 0], atol=TOL)
    np.testing.assert_allclose(img01, g[tag + "_img01"][:, 0], atol=TOL)
    tiled, tiled01, _ = _images(eng, ev, z, False)
    # (the other path did run; a process started with GDPSO_GEN_RESIDENT=0 packs generators without the resident form)
    assert not np.array_equal(img, tiled) or not ENV_GEN
    np.testing.assert_allclose(img, tiled, atol=2e-5)


@pytest.mark.parametrize("zd", [10, 100, 16, 3])
@pytest.mark.parametrize("n", [1, 2, 37, 256, 257, 512, 700, 1001, 2048])
def test_matches_tiled_kernels_and_oracle(eng, synth, oracle, zd, n):
    """One image a workgroup (up to one per CU), two (beyond), an odd batch whose last workgroup holds one image, a
    short last round in a launch of its own with one image a workgroup (700 on 256 CUs: 512 + 188);
    float64 latents like the reference's numpy positions, float32 ones as well."""
    ev = _evaluator(eng, synth, zd)
    zh = np.random.RandomState(100 * zd + n).standard_normal((n, zd))
    z = torch.from_numpy(zh).cuda()
    img, img01, _ = _images(eng, ev, z, True)
    tiled, tiled01, _ = _images(eng, ev, z, False)
    assert np.isfinite(img).all() and np.isfinite(img01).all()
    np.testing.assert_allclose(img, tiled, atol=2e-5)
    np.testing.assert_allclose(img01, tiled01, atol=2e-5)
    img32, _, _ = _images(eng, ev, z.float(), True)
    np.testing.assert_allclose(img32, img, atol=2e-5)
    gsd = synth.to_torch(synth.generator_state(zd, 1, 64, seed=11))
    sel = np.arange(n) if n <= 37 else np.concatenate([[0, 1, n - 2, n - 1], np.random.RandomState(1).choice(n, 12, replace=False)])
    with torch.no_grad():
        want = oracle.generator_forward(gsd, torch.from_numpy(zh[sel]).float().reshape(len(sel), zd, 1, 1))
        want01 = oracle.rescale_per_image(want)
    np.testing.assert_allclose(img[sel], want[:, 0].numpy(), atol=TOL)
    np.testing.assert_allclose(img01[sel], want01[:, 0].numpy(), atol=TOL)


@pytest.mark.parametrize("n", [300, 700])
@pytest.mark.parametrize("zd", [10, 100])
def test_inverter_error_of_the_launch(eng, synth, zd, n):
    """The mean squared error of each image against its target (the inverter's fitness term, reference
    src/utils/util_discovery.py:97) comes out of the same launch (700 images on 256 CUs: the last 188 from a second
    launch with one image a workgroup, which reads its own targets and writes its own errors)."""
    ev = _evaluator(eng, synth, zd)
    z = torch.from_numpy(np.random.RandomState(5).standard_normal((n, zd))).cuda()
    target = torch.from_numpy((2 * np.random.RandomState(6).random_sample((n, 28, 28)) - 1).astype(np.float32)).cuda()
    img, img01, mse = _images(eng, ev, z, True, target)
    _, tiled01, tiled_mse = _images(eng, ev, z, False, target)
    np.testing.assert_allclose(mse, tiled_mse, rtol=1e-5, atol=1e-7)
    want = ((img.astype(np.float64) - target.cpu().numpy().astype(np.float64)) ** 2).mean(axis=(1, 2))
    np.testing.assert_allclose(mse, want, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_one_and_two_images_a_workgroup_give_the_same_bits(eng, lib, synth, dtype):
    if not ENV_GEN:
        pytest.skip("generators are packed without the resident form in this process (the tiled kernels' sums depend on the batch)")
    ev = _evaluator(eng, synth, 10, dtype)
    z = torch.from_numpy(np.random.RandomState(9).standard_normal((161, 10))).cuda()
    out = []
    for images in (1, 2):
        lib.check(lib.lib().gdpso_resident_force_images(images))
        try:
            out.append(_images(eng, ev, z, True))
        finally:
            lib.check(lib.lib().gdpso_resident_force_images(0))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    # an image's bits do not depend on its place in the batch either
    odd = _images(eng, ev, z[1:66].contiguous(), True)
    np.testing.assert_array_equal(odd[0], out[0][0][1:66])


@pytest.mark.parametrize("dtype,bar", [("bf16", 0.05), ("f16", 0.01)])
@pytest.mark.parametrize("zd", [10, 100])
def test_16bit_storage(eng, synth, oracle, dtype, bar, zd):
    """16-bit maps and weights: each path rounds every layer's result to the element type; both are held to the
    16-bit configurations' bar on the image (|d| against the fp32 oracle) and to that bar against each other."""
    ev = _evaluator(eng, synth, zd, dtype)
    n = 300
    zh = np.random.RandomState(17).standard_normal((n, zd))
    z = torch.from_numpy(zh).cuda()
    img, img01, _ = _images(eng, ev, z, True)
    tiled, tiled01, _ = _images(eng, ev, z, False)
    np.testing.assert_allclose(img, tiled, atol=bar)
    gsd = synth.to_torch(synth.generator_state(zd, 1, 64, seed=11))
    sel = np.random.RandomState(1).choice(n, 16, replace=False)
    with torch.no_grad():
        want = oracle.generator_forward(gsd, torch.from_numpy(zh[sel]).float().reshape(len(sel), zd, 1, 1))[:, 0].numpy()
    np.testing.assert_allclose(img[sel], want, atol=bar)
    assert np.abs(img[sel] - want).mean() <= 1.5 * np.abs(tiled[sel] - want).mean() + 1e-4


def test_planner_keeps_fp32_batches_that_leave_cus_idle_on_the_tiled_kernels(eng, synth):
    """Mode 1 (the default): in fp32 the one launch is bound by the matrix pipe of the CUs that got a workgroup, so a
    batch of half an image per CU stays with the tiled kernels and one of two images per CU takes the launch; with
    16-bit storage the launch is taken either way."""
    if not ENV_GEN:
        pytest.skip("generators are packed without the resident form in this process")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    ev = _evaluator(eng, synth, 10)
    rs = np.random.RandomState(23)
    small, full = (torch.from_numpy(rs.standard_normal((n, 10))).cuda() for n in (cus // 2, 2 * cus))
    np.testing.assert_array_equal(_images(eng, ev, small, 1, min_images=64)[0], _images(eng, ev, small, 0, min_images=64)[0])
    np.testing.assert_array_equal(_images(eng, ev, full, 1, min_images=64)[0], _images(eng, ev, full, 2, min_images=64)[0])
    assert not np.array_equal(_images(eng, ev, full, 1, min_images=64)[0], _images(eng, ev, full, 0, min_images=64)[0])
    ev16 = _evaluator(eng, synth, 10, "bf16")
    # (with 16-bit storage the two paths' images can agree bit for bit: a layer's rounding hides the order of its sums)
    np.testing.assert_array_equal(_images(eng, ev16, small, 1, min_images=64)[0], _images(eng, ev16, small, 2, min_images=64)[0])


def test_limits_and_other_shapes_keep_the_tiled_kernels(eng, synth):
    """Below the resident launches' lower limit the tiled kernels run (same bits as with the switch off); a 64x64
    generator (first kernel 16) has no resident form."""
    ev = _evaluator(eng, synth, 10)
    z = torch.from_numpy(np.random.RandomState(3).standard_normal((48, 10))).cuda()
    np.testing.assert_array_equal(_images(eng, ev, z, True, min_images=64)[0], _images(eng, ev, z, False)[0])
    dev = torch.device("cuda:0")
    gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11, first_kernel=16), dev)
    ev64 = eng.Evaluator(gen, eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=64))
    np.testing.assert_array_equal(_images(eng, ev64, z, True)[0], _images(eng, ev64, z, False)[0])
