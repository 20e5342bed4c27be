"""GPU parity of the network kernels against the reference's own activations
(tests/golden/layers.npz, assessor_shapes.npz) -- through the C ABI."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden

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
