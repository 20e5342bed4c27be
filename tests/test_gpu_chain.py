"""The chained launch (every convolution of the ResNet stages behind the first in one persistent
launch, intermediates handed over inside an XCD's L2) against the per-layer launches of the same
plan -- bit for bit, the tile code and the summation order are the same -- and against the oracle;
with an even deal of workgroups to XCDs and with XCDs standing aside (the uneven-deal path)."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest
import torch

# The development switch that takes the small tiles away (tools/switch_matrix.sh) changes which layers
# can be chained and makes the per-layer plan split K in the late stages: these comparisons assume the
# default tile set.
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("GDPSO_CONV_TILE32", "1") == "0", reason="assumes the default tile set")]
TOL = 1e-4


@pytest.fixture(scope="module")
def eng():
    return importlib.import_module("gan-discovery-pso_amd.engine")


@pytest.fixture(scope="module")
def lib():
    return importlib.import_module("gan-discovery-pso_amd._lib")


def _stats(lib):
    launches, aborts = C.c_uint(0), C.c_uint(0)
    lib.check(lib.lib().gdpso_chain_stats(C.byref(launches), C.byref(aborts)))
    return launches.value, aborts.value


def _aborts(lib):
    return _stats(lib)[1]


def _logits(eng, lib, synth, model, n, side, chained, drop=0, seed=5, teams=1, dtype="f32"):
    """Logits of `n` seeded images through a fresh assessor whose plans are made with the switch set."""
    dev = torch.device("cuda:0")
    lib.check(lib.lib().gdpso_chain_debug(int(chained), drop, teams))
    try:
        ass = eng.PackedAssessor(synth.resnet_state(model, 1, 8, seed=12), dev, img_side=side, dtype=dtype)
        gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11, first_kernel=side // 4), dev, dtype=dtype)
        ev = eng.Evaluator(gen, ass)
        x = torch.from_numpy(np.random.RandomState(seed).random_sample((n, side, side)).astype(np.float32)).cuda()
        out = []
        launched = _stats(lib)[0]
        for _ in range(3):  # the same buffers again: stale cache lines would show on the second pass
            _, logits = ev.assess(x, eng.FitnessSpec(class_index=0), want_logits=True)
            out.append(logits.cpu().numpy().copy())
        torch.cuda.synchronize()
        assert _stats(lib)[0] - launched == (3 if chained else 0)  # the path under test is the one that ran
    finally:
        lib.check(lib.lib().gdpso_chain_debug(1, 0, 1))
    for o in out[1:]:
        np.testing.assert_array_equal(o, out[0])
    return out[0], x


@pytest.mark.parametrize("model,n,side,exact", [("ResNet50", 512, 28, True), ("ResNet50", 256, 28, False), ("ResNet50", 1024, 28, True),
                                                 ("ResNet101", 512, 28, True), ("ResNet50", 768, 28, False), ("ResNet152", 512, 28, True),
                                                 ("ResNet50", 300, 28, False), ("ResNet50", 500, 28, False), ("ResNet50", 1001, 28, False),
                                                 ("ResNet50", 1, 28, False), ("ResNet50", 5, 28, False), ("ResNet50", 37, 28, False),
                                                 ("ResNet50", 100, 64, False)])
def test_chain_matches_per_layer_launches(eng, lib, synth, model, n, side, exact):
    """768 images: 96 a share, so the last 64-row tile of a share is half empty; 300, 500, 1001: shares of 38, 63 and
    126 images, the last share shorter (1001: much shorter), partly filled tiles in most phases; 1 and 5 images: most
    shares empty; 100 images of 64x64: the larger maps.
    exact: the per-layer plan has no split-K layer at this size, so both paths add in the same order (at
    256 images the late stages alone have fewer tiles than the device has CUs and are split; a chain never is)."""
    before = _aborts(lib)
    plain, _ = _logits(eng, lib, synth, model, n, side, chained=False)
    chained, _ = _logits(eng, lib, synth, model, n, side, chained=True)
    if exact:
        np.testing.assert_array_equal(chained, plain)
    else:
        np.testing.assert_allclose(chained, plain, rtol=1e-4, atol=2e-4)
    assert _aborts(lib) == before


@pytest.mark.parametrize("n,side,drop,teams", [(57, 64, 0, 1), (57, 64, 0xF0, 1), (57, 64, 0, 2), (9, 64, 0, 1), (41, 64, 0x7F, 1), (121, 28, 0, 1),
                                               (57, 64, 0xFE, 2)])
def test_chain_uneven_batches_on_larger_maps(eng, lib, synth, n, side, drop, teams):
    """Batches that do not divide by the number of shares, on maps large enough for a share to have MORE tiles
    in a step than its team has members: a member's first tile of a step may then lie in the empty tail of one
    border class (past the last share's rows) while a later tile of the same member is a real one in the next
    class -- the step's wait must have been taken by then.  57 images of 64x64: 8 a share and 1 in the last, 84
    tiles of 32x32 in the 3x3 layers of the second stage for 64 members.  With XCDs standing aside the remaining
    ones walk several shares each.  K is never split in either path here, so both add in the same order."""
    before = _aborts(lib)
    eng.plan_options(plan_images=0, splitk=False)
    try:
        plain, _ = _logits(eng, lib, synth, "ResNet50", n, side, chained=False)
        chained, _ = _logits(eng, lib, synth, "ResNet50", n, side, chained=True, drop=drop, teams=teams)
    finally:
        eng.plan_options(plan_images=0, splitk=True)
    if teams == 1:
        np.testing.assert_array_equal(chained, plain)
    else:  # two teams: shares of a sixteenth, some layers on smaller tiles whose K parts add in another order
        np.testing.assert_allclose(chained, plain, rtol=1e-4, atol=2e-4)
    assert _aborts(lib) == before


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_chain_16bit_matches_per_layer_launches(eng, lib, synth, dtype):
    """16-bit storage: the chained launch runs the same tiles in the same order as the launches per layer."""
    before = _aborts(lib)
    plain, _ = _logits(eng, lib, synth, "ResNet50", 512, 28, chained=False, dtype=dtype)
    chained, _ = _logits(eng, lib, synth, "ResNet50", 512, 28, chained=True, dtype=dtype)
    np.testing.assert_array_equal(chained, plain)
    assert _aborts(lib) == before


@pytest.mark.parametrize("drop", [0xF0, 0x0F, 0xFE, 0x24])
def test_chain_with_xcds_standing_aside(eng, lib, synth, drop):
    before = _aborts(lib)
    plain, _ = _logits(eng, lib, synth, "ResNet50", 512, 28, chained=False)
    chained, _ = _logits(eng, lib, synth, "ResNet50", 512, 28, chained=True, drop=drop)
    np.testing.assert_array_equal(chained, plain)
    assert _aborts(lib) == before


@pytest.mark.parametrize("drop", [0, 0xF0, 0x7F])
def test_two_teams_per_xcd(eng, lib, synth, drop):
    """Two teams per XCD (sixteen shares): some layers then run with smaller tiles, whose K parts are
    added in another order, so the comparison is to fp32 rounding, not bit for bit."""
    before = _aborts(lib)
    plain, _ = _logits(eng, lib, synth, "ResNet50", 512, 28, chained=False)
    chained, _ = _logits(eng, lib, synth, "ResNet50", 512, 28, chained=True, drop=drop, teams=2)
    np.testing.assert_allclose(chained, plain, rtol=1e-4, atol=2e-4)
    assert _aborts(lib) == before


def test_chain_against_oracle(eng, lib, synth, oracle):
    n = 512
    got, x = _logits(eng, lib, synth, "ResNet50", n, 28, chained=True, seed=9)
    csd = synth.to_torch(synth.resnet_state("ResNet50", 1, 8, seed=12))
    sel = np.concatenate([[0, 63, 64, n - 1], np.random.RandomState(1).choice(n, 12, replace=False)])
    with torch.no_grad():
        want = oracle.resnet_forward(csd, torch.from_numpy(x.cpu().numpy()[sel][:, None])).numpy()
    np.testing.assert_allclose(got[sel], np.asarray(want), rtol=1e-4, atol=2e-4)


def test_swarm_run_is_the_same_chained(eng, lib, synth):
    """A whole 512-particle swarm run (Philox draws on the device) ends in the same state either way."""
    dev = torch.device("cuda:0")
    res = []
    for chained in (0, 1):
        lib.check(lib.lib().gdpso_chain_debug(chained, 0, 1))
        try:
            gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11), dev)
            ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=28)
            ev = eng.Evaluator(gen, ass)
            z = torch.from_numpy(np.random.RandomState(4).standard_normal((512, 10))).cuda()
            fit, _, _ = ev.fitness(z, eng.FitnessSpec(class_index=3))
            res.append(fit.cpu().numpy().copy())
        finally:
            lib.check(lib.lib().gdpso_chain_debug(1, 0, 1))
    np.testing.assert_array_equal(res[1], res[0])


def test_chains_on_two_streams(eng, lib, synth):
    """Two evaluators on two streams, launches interleaved from the host: once the library sees a second
    stream on the device it stops chaining there (a chain's workgroups hold their compute units while they
    wait for each other: a bad neighbour for another stream's kernels); results are those of either
    evaluator alone."""
    dev = torch.device("cuda:0")
    lib.check(lib.lib().gdpso_chain_debug(1, 0, 1))
    before = _aborts(lib)
    gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11), dev)
    ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=28)
    evs = [eng.Evaluator(gen, ass), eng.Evaluator(gen, ass)]
    z = [torch.from_numpy(np.random.RandomState(s).standard_normal((512, 10))).cuda() for s in (1, 2)]
    spec = eng.FitnessSpec(class_index=2)
    want = [evs[i].fitness(z[i], spec)[0].cpu().numpy().copy() for i in range(2)]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    torch.cuda.synchronize()
    got = [None, None]
    for rep in range(3):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                got[i] = evs[i].fitness(z[i], spec)[0]
    torch.cuda.synchronize()
    launched = _stats(lib)[0]
    with torch.cuda.stream(streams[0]):
        evs[0].fitness(z[0], spec)
    torch.cuda.synchronize()
    assert _stats(lib)[0] == launched  # a device that has seen two streams runs one launch per layer
    lib.check(lib.lib().gdpso_chain_debug(1, 0, 1))  # forget the streams: the tests behind this one chain again
    for i in range(2):
        np.testing.assert_array_equal(got[i].cpu().numpy(), want[i])
    assert _aborts(lib) == before
