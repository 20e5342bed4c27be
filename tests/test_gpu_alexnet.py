"""The small "AlexNet" assessor on the GPU (reference src/pso/util_cnn.py:193-249) against outputs of
the reference's own class (tests/golden/alexnet.npz) -- through get_cnn / Discovery / Swarm."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from parity_utils import first_ambiguous_iteration

pytestmark = pytest.mark.gpu
TOL = 1e-4
imp = lambda m: importlib.import_module("gan-discovery-pso_amd." + m)
ARGS28 = {"kernel": 3, "padding": "same", "cnn_activation": "LeakyReLU"}
ARGS64 = {"kernel": 3, "padding": "valid", "cnn_activation": "ReLU"}


def make(synth, side, args, seed, dev):
    iid = [0, 2, 3, 4, 6, 7, 8, 9]
    cnn = imp("pso.util_cnn").get_cnn("AlexNet", 1, iid, 8, img_dim=side, cnn_args=args).to(dev)
    cnn.load_state_dict(synth.to_torch(synth.alexnet_state(8, kernel=3, img_size=side, padding=args["padding"], seed=seed)))
    return cnn


def test_logits_match_reference(synth):
    g = load_golden("alexnet")
    dev = torch.device("cuda:0")
    cnn28 = make(synth, 28, ARGS28, int(g["seed28"]), dev)
    np.testing.assert_allclose(cnn28(torch.from_numpy(g["x28"]).to(dev)).cpu().numpy(), g["logits28"], rtol=1e-4, atol=TOL)
    cnn64 = make(synth, 64, ARGS64, int(g["seed64"]), dev)
    assert cnn64._to_linear == int(g["to_linear64"])
    np.testing.assert_allclose(cnn64(torch.from_numpy(g["x64"]).to(dev)).cpu().numpy(), g["logits64"], rtol=1e-4, atol=TOL)
    # larger batches (row tails, other tile shapes) against the oracle on the same weights
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pso_oracle as oracle
    x = np.random.RandomState(4).uniform(0, 1, (70, 1, 28, 28)).astype(np.float32)
    sd = synth.to_torch(synth.alexnet_state(8, kernel=3, img_size=28, padding="same", seed=int(g["seed28"])))
    with torch.no_grad():
        want = oracle.alexnet_forward(sd, torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(cnn28(torch.from_numpy(x).to(dev)).cpu().numpy(), want, rtol=1e-4, atol=TOL)


def test_constructor_fails_like_the_reference(synth):
    """padding 'valid' at 28x28 (what configs/dcgan_mnist.yaml:118-120 would give): the 4th convolution
    sees a 1x1 map; the reference's constructor raises RuntimeError from its dummy forward pass."""
    g = load_golden("alexnet")
    with pytest.raises(RuntimeError, match="Kernel size can't be greater than actual input size"):
        imp("pso.util_cnn").get_cnn("AlexNet", 1, [0, 1], 2, img_dim=28, cnn_args={"kernel": 3, "padding": "valid", "cnn_activation": "ReLU"})
    assert str(g["valid28_error"]) == "RuntimeError"
    assert imp("pso.util_cnn").get_cnn("VGG16", 1, [0, 1], 2) is None
    with pytest.raises(ValueError):
        imp("pso.util_cnn").get_cnn("LeNet", 1, [0, 1], 2)


def test_fitness_and_swarm_with_alexnet(synth):
    g = load_golden("alexnet")
    dev = torch.device("cuda:0")
    cnn = make(synth, 28, ARGS28, int(g["seed28"]), dev)
    gen = imp("utils.util_dcgan").Generator(10, 1, 64).to(dev)
    gen.load_state_dict(synth.to_torch(synth.generator_state(10, 1, 64, seed=int(g["gen_seed"]))))
    disc = imp("pso.util_discovery").Discovery(int(g["label"]), gen, cnn, dev, iid_classes=g["iid"].tolist())
    np.testing.assert_allclose(disc.fitness_batch(g["z"]).cpu().numpy(), g["fitness"], atol=TOL)
    f, img01, img = disc.fitness(10, g["z"][2])
    np.testing.assert_allclose(f, g["fitness"][2:3], atol=TOL)
    n, t, d = int(g["sw_n"]), int(g["sw_t"]), int(g["sw_d"])
    np.random.seed(int(g["sw_np_seed"]))
    sw = imp("pso.util_pso").Swarm("/tmp/gdpso_test", disc, n, t, d, dev, tolerance=float(g["sw_tolerance"]),
                                   w_inertia=g["sw_w"][0], w_cogn=g["sw_w"][1], w_soci=g["sw_w"][2])
    sw.verbose, sw.record_fitness = False, True
    sw.optimize(schedule_inertia=False, early_stopping=False)
    res = sw.results
    safe = first_ambiguous_iteration(g["sw_fitness"], 2e-5)
    assert safe >= 3
    np.testing.assert_allclose(res["fitness"][:safe], g["sw_fitness"][:safe], atol=TOL)
    np.testing.assert_array_equal(res["hist_pos"][:safe + 1], g["sw_pos"][:safe + 1])
    np.testing.assert_allclose(res["g_best_val_dummy"][:safe], g["sw_g_best_val_dummy"][:safe], atol=TOL)


@pytest.mark.parametrize("dtype,bar", [("bf16", 0.12), ("f16", 0.025)])
def test_alexnet_16bit(synth, dtype, bar):
    """bf16 / fp16 storage for this assessor too (the first layer writes 64 channels, 32 of them
    zeros, for the 16-bit kernels' channel quantum): fitness within the stated bars of the reference."""
    g = load_golden("alexnet")
    dev = torch.device("cuda:0")
    eng = imp("engine")
    ass = eng.PackedAssessor(synth.alexnet_state(8, kernel=3, img_size=28, padding="same", seed=int(g["seed28"])), dev, img_side=28,
                             dtype=dtype, cnn_args=ARGS28)
    gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=int(g["gen_seed"])), dev, dtype=dtype)
    ci = sorted(g["iid"].tolist()).index(int(g["label"]))
    fit, _, _ = eng.Evaluator(gen, ass).fitness(torch.from_numpy(g["z"]).cuda(), eng.FitnessSpec(class_index=ci))
    np.testing.assert_allclose(fit.cpu().numpy(), g["fitness"], atol=bar)


def test_state_dict_geometry_is_checked_against_fc1(synth):
    """A plain state dict carries no cnn_args: the padding is read off fc1's width (which the geometry fixes),
    a padding that contradicts it is refused, and a missing activation is assumed with a warning."""
    g = load_golden("alexnet")
    dev = torch.device("cuda:0")
    eng = imp("engine")
    sd64 = synth.alexnet_state(8, kernel=3, img_size=64, padding="valid", seed=int(g["seed64"]))
    assert eng.alexnet_to_linear(64, 3, "valid") == int(g["to_linear64"]) == sd64["fc1.weight"].shape[1]
    assert eng.alexnet_to_linear(28, 3, "valid") is None and eng.alexnet_to_linear(28, 3, "same") == 256
    x = torch.from_numpy(g["x64"]).to(dev)
    want = g["logits64"]
    cnn_mod = imp("pso.util_cnn")
    # padding inferred ('valid': 'same' would give another fc1 width), activation given
    ass = eng.PackedAssessor(sd64, dev, img_side=64, cnn_args={"cnn_activation": "ReLU"})
    np.testing.assert_allclose(cnn_mod._forward_logits(ass, x).cpu().numpy(), want, rtol=1e-4, atol=TOL)
    with pytest.raises(ValueError, match="fc1 takes"):
        eng.PackedAssessor(sd64, dev, img_side=64, cnn_args={"padding": "same", "cnn_activation": "ReLU"})
    with pytest.raises(ValueError, match="neither padding"):
        eng.PackedAssessor(sd64, dev, img_side=28, cnn_args={"cnn_activation": "ReLU"})
    with pytest.warns(UserWarning, match="activation"):
        eng.PackedAssessor(sd64, dev, img_side=64)
