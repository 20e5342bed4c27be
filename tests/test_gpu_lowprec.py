"""The reduced-precision configurations on the GPU: activations and convolution weights stored as
bf16 (VQ-VAE decoder as generator, Z = 100: configs/vqvae.yaml) or fp16 (the 64x64 generator), 16-bit
matrix instructions with fp32 accumulation and an fp32 epilogue.

The reference has no reduced-precision mode (it is fp32 throughout), so these paths are judged against
its fp32 outputs with a STATED tolerance, measured on the fixtures and on 256 random latents per
network (tools/probe16.py):

                       images |d|   logits |d| (scale 8)   fitness |d|  max / mean
    bf16 (8-bit mantissa)   1.0e-2        0.25                  0.074 / 0.017
    fp16 (11-bit mantissa)  1.4e-3        0.026                 0.011 / 0.004

The bars below are those maxima with a margin.  Decisions of a swarm run on such fitness values differ
from the fp32 trajectory as soon as two candidates are closer than the error, so a 16-bit swarm is
checked for what must hold whatever the precision: the loop's arithmetic applied to the fitness
values the run itself produced reproduces its positions bit for bit (oracle loop with injected
fitness), and every iteration's fitness is within the bar of the fp32 oracle at the same positions."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
# absolute bars against the fp32 reference: (image, logits, fitness max, fitness mean)
BARS = {"bf16": (2.5e-2, 0.6, 0.12, 0.04), "f16": (4e-3, 0.08, 0.025, 0.01)}


@pytest.fixture(scope="module")
def eng():
    return importlib.import_module("gan-discovery-pso_amd.engine")


def make_eval(eng, synth, zd, dtype, side=28, gseed=11, cseed=12):
    dev = torch.device("cuda:0")
    gen = eng.PackedGenerator(synth.generator_state(zd, 1, 64, seed=gseed, first_kernel=side // 4), dev, dtype=dtype)
    ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=cseed), dev, img_side=side, dtype=dtype)
    return eng.Evaluator(gen, ass)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("tag,zd", [("z10", 10), ("z100", 100)])
def test_networks_against_reference_activations(eng, synth, dtype, tag, zd):
    """layers.npz (outputs of the reference's own Generator / Resnet / Discovery) within the bars."""
    g = load_golden("layers")
    b_img, b_log, b_fit, _ = BARS[dtype]
    ev = make_eval(eng, synth, zd, dtype)
    z = torch.from_numpy(g[tag + "_z"]).cuda()
    img01, img, _ = ev.images(z)
    np.testing.assert_allclose(img.cpu().numpy(), g[tag + "_img"][:, 0], atol=b_img)
    np.testing.assert_allclose(img01.cpu().numpy(), g[tag + "_img01"][:, 0], atol=b_img)
    assert float(img01.min()) == 0.0 and float(img01.max()) == 1.0  # the rescale itself is fp32
    ci = sorted(g[tag + "_iid"].tolist()).index(int(g[tag + "_label"]))
    fit, logits = ev.assess(img01, eng.FitnessSpec(class_index=ci), want_logits=True)
    np.testing.assert_allclose(logits.cpu().numpy(), g[tag + "_logits"], atol=b_log)
    np.testing.assert_allclose(fit.cpu().numpy(), g[tag + "_fitness"], atol=b_fit)


@pytest.mark.parametrize("dtype,zd,side", [("bf16", 100, 28), ("f16", 10, 64), ("bf16", 10, 28), ("f16", 100, 28)])
def test_fitness_against_fp32_oracle_on_many_latents(eng, oracle, synth, dtype, zd, side):
    """200 random latents (a batch that is no multiple of any tile) through the fp32 oracle and
    through the 16-bit networks: maximum and mean |difference| of the fitness within the bars.
    ("bf16", 100, 28) is the VQ-VAE-decoder configuration, ("f16", 10, 64) the 64x64 one.)"""
    _, _, b_fit, b_mean = BARS[dtype]
    ev = make_eval(eng, synth, zd, dtype, side)
    gsd = synth.to_torch(synth.generator_state(zd, 1, 64, seed=11, first_kernel=side // 4))
    csd = synth.to_torch(synth.resnet_state("ResNet50", 1, 8, seed=12))
    n = 200 if side == 28 else 60
    z = np.random.RandomState(31).standard_normal((n, zd))
    want = oracle.Fitness(gsd, csd, 6)(z)
    got, _, _ = ev.fitness(torch.from_numpy(z).cuda(), eng.FitnessSpec(class_index=6))
    d = np.abs(got.cpu().numpy() - want)
    assert d.max() <= b_fit and d.mean() <= b_mean, (d.max(), d.mean())


@pytest.mark.parametrize("dtype,name", [("bf16", "swarm_n16_t30_z100_sched_es"), ("f16", "swarm_n64_t50_z10")])
def test_swarm_run_is_self_consistent(eng, oracle, synth, dtype, name):
    """A whole swarm run on 16-bit networks from a fixture's initial state and draws: (a) the oracle's
    loop fed the fitness values the run recorded reproduces every position, velocity and best bit for
    bit (the swarm state stays float64 and the decisions follow the device's fitness exactly); (b) at
    every iteration the recorded fitness is within the bar of the fp32 oracle at the run's positions."""
    g = load_golden(name)
    n, t, d = int(g["n"]), int(g["t"]), int(g["d"])
    _, _, b_fit, _ = BARS[dtype]
    ev = make_eval(eng, synth, d, dtype, gseed=int(g["gen_seed"]), cseed=int(g["cnn_seed"]))
    ci = sorted(g["iid"].tolist()).index(int(g["label"]))
    sched, es, tol = bool(g["schedule_inertia"]), bool(g["early_stopping"]), float(g["tolerance"])
    rand = np.random.RandomState(9).random_sample((t, n, 2))
    sw = eng.HipSwarm(ev, n, d, t, record_fitness=True)
    sw.load(g["pos"][0], g["vel"][0], float(g["w"][0]), rand=rand)
    eng.run_loop(sw, t, eng.FitnessSpec(class_index=ci), float(g["w"][1]), float(g["w"][2]), schedule_inertia=sched,
                 early_stopping=es, tolerance=tol)
    res = sw.results()
    iters = res["iters"]
    assert 3 <= iters <= t
    fit_rows = res["fitness"]

    def injected(pos, first):  # the run's own fitness, iteration by iteration
        injected.calls += 1
        return fit_rows[injected.calls - 1]

    injected.calls = 0
    ref = oracle.run_swarm(injected, g["pos"][0], g["vel"][0], t, float(g["w"][0]), float(g["w"][1]), float(g["w"][2]),
                           tolerance=tol, schedule_inertia=sched, early_stopping=es, rand=rand, batch=n)
    assert ref["i"] == (iters if res["stopped"] else t + 1)
    np.testing.assert_array_equal(res["hist_pos"], np.stack(ref["pos"]))
    np.testing.assert_array_equal(res["hist_vel"], np.stack(ref["vel"]))
    np.testing.assert_array_equal(res["g_best_val_dummy"], np.array(ref["g_best_val_dummy"], np.float32))
    np.testing.assert_array_equal(res["g_best_pos"], ref["g_best_pos"])
    np.testing.assert_allclose(res["mean_mse"], ref["mean_mse"], rtol=1e-12)
    gsd = synth.to_torch(synth.generator_state(d, 1, 64, seed=int(g["gen_seed"])))
    csd = synth.to_torch(synth.resnet_state("ResNet50", 1, 8, seed=int(g["cnn_seed"])))
    f32 = oracle.Fitness(gsd, csd, ci)
    for i in range(0, iters, max(1, iters // 6)):
        np.testing.assert_allclose(fit_rows[i], f32(res["hist_pos"][i]), atol=b_fit, err_msg=f"iteration {i + 1}")


def test_discovery_compute_dtype_knob(eng, synth):
    """Discovery.compute_dtype switches the packed networks (and re-packs when changed)."""
    imp = lambda m: importlib.import_module("gan-discovery-pso_amd." + m)
    dev = torch.device("cuda:0")
    gen = imp("utils.util_dcgan").Generator(10, 1, 64).to(dev)
    gen.load_state_dict(synth.to_torch(synth.generator_state(10, 1, 64, seed=11)))
    cnn = imp("pso.util_cnn").get_cnn("ResNet50", 1, list(range(8)), 8).to(dev)
    cnn.load_state_dict(synth.to_torch(synth.resnet_state("ResNet50", 1, 8, seed=12)))
    disc = imp("pso.util_discovery").Discovery(6, gen, cnn, dev, iid_classes=list(range(8)))
    z = np.random.RandomState(2).standard_normal((32, 10))
    f32 = disc.fitness_batch(z).cpu().numpy()
    disc.compute_dtype = "bf16"
    assert disc.evaluator().gen.dtype == "bf16"
    b16 = disc.fitness_batch(z).cpu().numpy()
    assert 0 < np.abs(b16 - f32).max() <= BARS["bf16"][2]
    disc.compute_dtype = "f32"
    np.testing.assert_array_equal(disc.fitness_batch(z).cpu().numpy(), f32)
    with pytest.raises(ValueError):
        eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11), dev, dtype="int8")
