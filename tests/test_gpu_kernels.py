"""GPU parity of the network kernels against the reference's own activations
(tests/golden/layers.npz, assessor_shapes.npz) -- through the C ABI."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from parity_utils import first_ambiguous_iteration

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def eng():
    return importlib.import_module("gan-discovery-pso_amd.engine")


@pytest.fixture(scope="module")
def nets(eng, synth):
    dev = torch.device("cuda:0")
    out = {}
    for zd in (10, 100):
        gen = eng.PackedGenerator(synth.generator_state(zd, 1, 64, seed=11), dev)
        ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=28)
        out[zd] = eng.Evaluator(gen, ass)
    return out


@pytest.mark.parametrize("tag,zd", [("z10", 10), ("z100", 100)])
def test_generator_and_fitness_match_reference(eng, nets, tag, zd):
    g = load_golden("layers")
    ev = nets[zd]
    z = torch.from_numpy(g[tag + "_z"]).cuda()  # float64, like the reference's numpy positions
    img01, img, _ = ev.images(z)
    np.testing.assert_allclose(img.cpu().numpy(), g[tag + "_img"][:, 0], atol=TOL)
    np.testing.assert_allclose(img01.cpu().numpy(), g[tag + "_img01"][:, 0], atol=TOL)
    ci = sorted(g[tag + "_iid"].tolist()).index(int(g[tag + "_label"]))
    spec = eng.FitnessSpec(class_index=ci, mode="optimize_out_training", threshold=0.0, eps=0.1)
    fit, logits = ev.assess(img01, spec, want_logits=True)
    np.testing.assert_allclose(logits.cpu().numpy(), g[tag + "_logits"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(fit.cpu().numpy(), g[tag + "_fitness"], atol=TOL)


def test_assessor_on_reference_images(eng, nets):
    """Feed the reference's own rescaled images: isolates the assessor from the generator."""
    g = load_golden("layers")
    ev = nets[10]
    x = torch.from_numpy(g["z10_img01"][:, 0].copy()).cuda()
    _, logits = ev.assess(x, eng.FitnessSpec(class_index=0), want_logits=True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["z10_logits"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("model", ["ResNet50", "ResNet101"])
@pytest.mark.parametrize("side", [28, 64])
def test_assessor_shapes(eng, synth, model, side):
    g = load_golden("assessor_shapes")
    dev = torch.device("cuda:0")
    ass = eng.PackedAssessor(synth.resnet_state(model, 1, 8, seed=12), dev, img_side=side)
    gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11, first_kernel=side // 4), dev)
    ev = eng.Evaluator(gen, ass)
    x = torch.from_numpy(np.ascontiguousarray(g["x64"][:, 0, :side, :side])).cuda()
    _, logits = ev.assess(x, eng.FitnessSpec(class_index=0), want_logits=True)
    np.testing.assert_allclose(logits.cpu().numpy(), g[f"{model}_logits{side}"], rtol=1e-4, atol=2e-4)


def test_batch_tail_and_large_batch(eng, nets, oracle, synth):
    """Batch sizes that are not a multiple of any tile (M tail rows, partly filled phases, every
    tile shape and launch form the planner picks along the way) and a 512 batch agree with the
    oracle on the same seeded latents."""
    ev = nets[10]
    gsd = synth.to_torch(synth.generator_state(10, 1, 64, seed=11))
    csd = synth.to_torch(synth.resnet_state("ResNet50", 1, 8, seed=12))
    fit = oracle.Fitness(gsd, csd, 6)
    spec = eng.FitnessSpec(class_index=6)
    rs = np.random.RandomState(3)
    for n in (1, 3, 37, 100, 257, 300, 512, 1000):
        z = rs.standard_normal((n, 10))
        got, _, _ = ev.fitness(torch.from_numpy(z).cuda(), spec)
        sel = np.arange(n) if n <= 37 else np.concatenate([[0, n - 1], rs.choice(n, 14, replace=False)])
        np.testing.assert_allclose(got.cpu().numpy()[sel], fit(z[sel]), atol=TOL)


@pytest.mark.parametrize("model", ["ResNet101", "ResNet152"])
def test_deep_assessors_against_oracle(eng, oracle, synth, model):
    """The 101- and 152-layer assessors (reference util_cnn.py:105-116) on generated images, an
    odd batch size, against the oracle's restatement of the same weights."""
    dev = torch.device("cuda:0")
    gstate = synth.generator_state(10, 1, 64, seed=21)
    cstate = synth.resnet_state(model, 1, 8, seed=22)
    ev = eng.Evaluator(eng.PackedGenerator(gstate, dev), eng.PackedAssessor(cstate, dev, img_side=28))
    fit = oracle.Fitness(synth.to_torch(gstate), synth.to_torch(cstate), 2)
    z = np.random.RandomState(8).standard_normal((11, 10))
    got, _, _ = ev.fitness(torch.from_numpy(z).cuda(), eng.FitnessSpec(class_index=2))
    np.testing.assert_allclose(got.cpu().numpy(), fit(z), atol=TOL)


# The 64x64 generator (first ConvTranspose kernel 16: 16 -> 32 -> 64) is this build's extension of the
# reference generator, which is hard-wired to 7 -> 14 -> 28 (reference src/utils/util_dcgan.py:135-137).
# PARITY UNPINNED BY THE REFERENCE: there is no reference output for it; it is checked against the
# oracle's restatement of the same three layers, which reads the first kernel size off the weights.
@pytest.fixture(scope="module")
def nets64(eng, synth):
    dev = torch.device("cuda:0")
    gstate = synth.generator_state(10, 1, 64, seed=11, first_kernel=16)
    cstate = synth.resnet_state("ResNet50", 1, 8, seed=12)
    ev = eng.Evaluator(eng.PackedGenerator(gstate, dev), eng.PackedAssessor(cstate, dev, img_side=64))
    return ev, synth.to_torch(gstate), synth.to_torch(cstate)


@pytest.mark.parametrize("n", [1, 5, 64])
def test_generator_64x64_against_oracle(eng, oracle, nets64, n):
    """Images, rescaled images and fitness of the 64x64 configuration (the last generator layer
    takes its unstaged VALU form here: neither the staged image nor the MFMA form fits LDS)."""
    ev, gsd, csd = nets64
    assert ev.gen.side == 64
    z = np.random.RandomState(40 + n).standard_normal((n, 10))
    img01, img, _ = ev.images(torch.from_numpy(z).cuda())
    with torch.no_grad():
        want = oracle.generator_forward(gsd, torch.from_numpy(z).float().reshape(n, 10, 1, 1))
        want01 = oracle.rescale_per_image(want)
    np.testing.assert_allclose(img.cpu().numpy(), want[:, 0].numpy(), atol=TOL)
    np.testing.assert_allclose(img01.cpu().numpy(), want01[:, 0].numpy(), atol=TOL)
    fit = oracle.Fitness(gsd, csd, 6)
    got, _, _ = ev.fitness(torch.from_numpy(z).cuda(), eng.FitnessSpec(class_index=6))
    np.testing.assert_allclose(got.cpu().numpy(), fit(z), atol=TOL)


def test_swarm_64x64_against_oracle(eng, oracle, nets64):
    """A short swarm at 64x64 (16 particles, 5 iterations) against the oracle's loop on the same
    draws: per-iteration fitness within 1e-4 and, while the decisions agree, identical positions."""
    ev, gsd, csd = nets64
    n, t, d = 16, 5, 10
    rs = np.random.RandomState(77)
    pos, vel = oracle.init_swarm(n, d, rs)
    rand = rs.random_sample((t, n, 2))
    ref = oracle.run_swarm(oracle.Fitness(gsd, csd, 6), pos, vel, t, 0.73, 1.496, 1.496, early_stopping=False, rand=rand, batch=n)
    sw = eng.HipSwarm(ev, n, d, t, record_fitness=True)
    sw.load(pos, vel, 0.73, rand=rand)
    eng.run_loop(sw, t, eng.FitnessSpec(class_index=6), 1.496, 1.496, early_stopping=False, tolerance=1e-4)
    res = sw.results()
    assert res["iters"] == t
    # decisions are strict compares on fp32 fitness: the trajectories are compared while every
    # compare of the oracle's run has a margin above the fp32 noise of the fitness
    safe = first_ambiguous_iteration(np.stack(ref["fitness"]), 2e-5)
    assert safe >= 2, "choose another seed: the first iterations of this run are too close to call"
    for i in range(safe):
        np.testing.assert_allclose(res["fitness"][i], ref["fitness"][i], atol=TOL)
    np.testing.assert_array_equal(res["hist_pos"][:safe + 1], np.stack(ref["pos"])[:safe + 1])
    np.testing.assert_allclose(res["g_best_val_dummy"][:safe], np.array(ref["g_best_val_dummy"][:safe], np.float32), atol=TOL)
