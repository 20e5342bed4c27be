"""The resident stages (the stem and the first, or the first two, ResNet stages in one launch, two images'
maps in a workgroup's LDS) against the tiled convolution kernels of the same assessor and against the
oracle.  The paths add a row's products in different orders: the comparison between them is to fp32
rounding; an image's result must not depend on the batch it arrives in nor on its place in the batch."""
import importlib

import numpy as np
import pytest
import torch

import os

pytestmark = pytest.mark.gpu
# what the process started with (tools/switch_matrix.sh runs the suite under the environment switches): restored after every case
ENV_STAGES = 0 if os.environ.get("GDPSO_STAGE1_RESIDENT", "1") == "0" else 1 if os.environ.get("GDPSO_STAGE2_RESIDENT", "1") == "0" else 2
ENV_MAX = int(os.environ.get("GDPSO_STAGE1_RESIDENT_MAX", "-1"))  # (-1: the library's default)
ENV_MIN = int(os.environ.get("GDPSO_STAGE1_RESIDENT_MIN", "-1"))
# (the second stage's own lower limit is an environment switch only)
S2_MIN = int(os.environ.get("GDPSO_STAGE2_RESIDENT_MIN", "160"))


@pytest.fixture(scope="module")
def eng():
    return importlib.import_module("gan-discovery-pso_amd.engine")


def _logits(eng, synth, model, x, resident, max_images=1 << 30, n_class=8, dtype="f32", min_images=1):
    dev = torch.device("cuda:0")
    eng.resident_options(int(resident), max_images, min_images)
    try:
        ass = eng.PackedAssessor(synth.resnet_state(model, 1, n_class, seed=12), dev, img_side=28, dtype=dtype)
        gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11), dev, dtype=dtype)
        ev = eng.Evaluator(gen, ass)
        out = []
        for _ in range(2):
            _, logits = ev.assess(x, eng.FitnessSpec(class_index=0), want_logits=True)
            out.append(logits.cpu().numpy().copy())
    finally:
        eng.resident_options(ENV_STAGES, ENV_MAX, ENV_MIN)
    np.testing.assert_array_equal(out[1], out[0])
    return out[0]


def _images(n, seed=5):
    return torch.from_numpy(np.random.RandomState(seed).random_sample((n, 28, 28)).astype(np.float32)).cuda()


@pytest.mark.parametrize("model,n", [("ResNet50", 512), ("ResNet50", 1), ("ResNet50", 2), ("ResNet50", 37), ("ResNet50", 301),
                                     ("ResNet50", 192), ("ResNet50", 257), ("ResNet50", 1024), ("ResNet50", 1001), ("ResNet50", 2048), ("ResNet50", 3000), ("ResNet101", 128), ("ResNet152", 64),
                                     ("ResNet101", 200), ("ResNet152", 176), ("ResNet152", 320)])
def test_resident_stage_matches_tiled_kernels(eng, synth, model, n):
    """512, 1024, 2048, 3000: two images a workgroup (one, two, four, six rounds of workgroups); 257, 301, 1001: an odd batch, the last
    workgroup with one image; up to one image per CU: one image a workgroup; ResNet152 from 160 images on: a second
    stage of eight blocks in the launch."""
    x = _images(n)
    tiled = _logits(eng, synth, model, x, resident=0)
    first = _logits(eng, synth, model, x, resident=1)
    both = _logits(eng, synth, model, x, resident=2)
    for resident in (first, both):
        assert np.isfinite(resident).all()
        np.testing.assert_allclose(resident, tiled, rtol=1e-4, atol=2e-4)
        assert not np.array_equal(resident, tiled) or n < 8  # (the other path did run: the sums are ordered differently)
    # (the second stage rides along from 160 images on: GDPSO_STAGE2_RESIDENT_MIN)
    assert np.array_equal(first, both) == (n < S2_MIN)


@pytest.mark.parametrize("dtype,bar", [("bf16", 0.6), ("f16", 0.08)])
@pytest.mark.parametrize("n", [512, 37, 200])
def test_resident_stage_16bit(eng, synth, oracle, dtype, bar, n):
    """16-bit storage: both paths round every layer's result to the element type, from sums added in different
    orders, so they differ where a rounding falls the other way; each is held to the bar of the 16-bit
    configurations against the fp32 oracle (logits |d|, tests/test_gpu_lowprec.py), and to that bar against the other."""
    x = _images(n, seed=7)
    tiled = _logits(eng, synth, "ResNet50", x, resident=0, dtype=dtype)
    first = _logits(eng, synth, "ResNet50", x, resident=1, dtype=dtype)
    np.testing.assert_allclose(first, tiled, atol=bar)
    resident = _logits(eng, synth, "ResNet50", x, resident=2, dtype=dtype)
    assert np.isfinite(resident).all()
    np.testing.assert_allclose(resident, tiled, atol=bar)
    csd = synth.to_torch(synth.resnet_state("ResNet50", 1, 8, seed=12))
    sel = np.random.RandomState(1).choice(n, 12, replace=False)
    with torch.no_grad():
        want = np.asarray(oracle.resnet_forward(csd, torch.from_numpy(x.cpu().numpy()[sel][:, None])).numpy())
    np.testing.assert_allclose(resident[sel], want, atol=bar)
    # no worse than the tiled kernels against the reference arithmetic
    assert np.abs(resident[sel] - want).mean() <= 1.5 * np.abs(tiled[sel] - want).mean() + 1e-3


def test_resident_stage_against_oracle(eng, synth, oracle):
    n = 100
    x = _images(n, seed=9)
    got = _logits(eng, synth, "ResNet50", x, resident=2)
    csd = synth.to_torch(synth.resnet_state("ResNet50", 1, 8, seed=12))
    sel = np.concatenate([[0, 1, 2, n - 2, n - 1], np.random.RandomState(1).choice(n, 11, replace=False)])
    with torch.no_grad():
        want = oracle.resnet_forward(csd, torch.from_numpy(x.cpu().numpy()[sel][:, None])).numpy()
    np.testing.assert_allclose(got[sel], np.asarray(want), rtol=1e-4, atol=2e-4)


def test_an_image_gets_the_same_bits_in_any_batch(eng, synth):
    """Batches of 512 and 64 are planned alike behind the first stage only under a pinned plan; the resident stage
    itself gives a row the same bits wherever it sits (first or second image of a pair, any workgroup)."""
    eng.plan_options(plan_images=64, splitk=False)
    try:
        x = _images(129)
        whole = _logits(eng, synth, "ResNet50", x, resident=2)
        odd = _logits(eng, synth, "ResNet50", x[1:66].contiguous(), resident=2)  # image k of this batch is image k + 1 of the other
    finally:
        eng.plan_options(plan_images=0, splitk=True)
    np.testing.assert_array_equal(odd, whole[1:66])


def test_limits_send_batches_to_the_tiled_kernels(eng, synth):
    """Above max_images and below min_images the tiled kernels run."""
    x = _images(64)
    tiled = _logits(eng, synth, "ResNet50", x, resident=0)
    np.testing.assert_array_equal(_logits(eng, synth, "ResNet50", x, resident=2, max_images=32), tiled)
    np.testing.assert_array_equal(_logits(eng, synth, "ResNet50", x, resident=2, min_images=65), tiled)


@pytest.mark.parametrize("extra", [0, 1, 37])
def test_short_last_round_takes_one_image_a_workgroup(eng, synth, extra):
    """One and a half rounds of workgroups (768 images on 256 CUs, and a few more): whole rounds two images a
    workgroup, the rest one image a workgroup in a launch of its own behind it; the same bits as every workgroup
    with two images, or every workgroup with one."""
    if os.environ.get("GDPSO_RESIDENT_TAIL", "1") == "0":
        pytest.skip("the one-image last round is switched off in this process")
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    x = _images(3 * cus + extra)
    mixed = _logits(eng, synth, "ResNet50", x, resident=2)
    tiled = _logits(eng, synth, "ResNet50", x, resident=0)
    np.testing.assert_allclose(mixed, tiled, rtol=1e-4, atol=2e-4)
    assert not np.array_equal(mixed, tiled)
    for images in (1, 2):
        lib.check(lib.lib().gdpso_resident_force_images(images))
        try:
            np.testing.assert_array_equal(_logits(eng, synth, "ResNet50", x, resident=2), mixed)
        finally:
            lib.check(lib.lib().gdpso_resident_force_images(0))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_one_and_two_images_a_workgroup_give_the_same_bits(eng, synth, dtype):
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    x = _images(161)
    out = []
    for images in (1, 2):
        lib.check(lib.lib().gdpso_resident_force_images(images))
        try:
            out.append(_logits(eng, synth, "ResNet50", x, resident=2, dtype=dtype))
        finally:
            lib.check(lib.lib().gdpso_resident_force_images(0))
    np.testing.assert_array_equal(out[0], out[1])
