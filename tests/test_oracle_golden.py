"""Pin the CPU oracle against the fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from parity_utils import first_ambiguous_iteration, vq_margins

TOL = 1e-4  # the contract: fp32 outputs within 1e-4 of the reference CPU path


def _weights(synth, g):
    gsd = synth.generator_state(int(g["z_dim"]), 1, 64, seed=int(g["gen_seed"]))
    csd = synth.resnet_state("ResNet50", 1, int(g["n_class"]), seed=int(g["cnn_seed"]))
    if "gen_ck" in g.files:
        assert synth.checksum(gsd) == str(g["gen_ck"]) and synth.checksum(csd) == str(g["cnn_ck"])
    return synth.to_torch(gsd), synth.to_torch(csd)


def _class_index(g):
    return sorted(g["iid"].tolist()).index(int(g["label"])) if "iid" in g.files else 1


def test_layers(oracle, synth):
    g = load_golden("layers")
    for tag, zd in (("z10", 10), ("z100", 100)):
        gsd = synth.to_torch(synth.generator_state(zd, 1, 64, seed=int(g["gen_seed"])))
        csd = synth.to_torch(synth.resnet_state("ResNet50", 1, 8, seed=int(g["cnn_seed"])))
        z = torch.from_numpy(g[tag + "_z"]).float().reshape(4, zd, 1, 1)
        with torch.no_grad():
            img, (a1, a2) = oracle.generator_forward(gsd, z, stages=True)
            img01 = oracle.rescale_per_image(img)
            logits, st = oracle.resnet_forward(csd, img01, stages=True)
        for name, got in [("a1", a1), ("a2", a2), ("img", img), ("img01", img01), ("stem", st[0]), ("s1", st[1]),
                          ("s2", st[2]), ("s3", st[3]), ("s4", st[4]), ("logits", logits)]:
            ref = g[f"{tag}_{name}"]
            np.testing.assert_allclose(got.numpy(), ref, rtol=1e-4, atol=1e-4, err_msg=f"{tag} {name}")
        ci = sorted(g[tag + "_iid"].tolist()).index(int(g[tag + "_label"]))
        f = oracle.fitness_from_logits(logits, ci, "optimize_out_training", 0.0, 0.1).numpy()
        np.testing.assert_allclose(f, g[tag + "_fitness"], atol=TOL)


@pytest.mark.parametrize("name", ["swarm_n8_t5_z10", "swarm_n64_t50_z10", "swarm_n16_t30_z100_sched_es",
                                  "swarm_n12_t40_z10_es", "swarm_n64_t50_z10_margin"])
def test_swarm_trajectory(oracle, synth, name):
    """Same seeds in -> the oracle reproduces the reference's whole trajectory."""
    g = load_golden(name)
    gsd, csd = _weights(synth, g)
    fit = oracle.Fitness(gsd, csd, _class_index(g), str(g["mode"]))
    n, d = int(g["n"]), int(g["d"])
    np.random.seed(int(g["np_seed"]))
    pos, vel = oracle.init_swarm(n, d)
    np.testing.assert_array_equal(pos, g["pos"][0])
    np.testing.assert_array_equal(vel, g["vel"][0])
    w = g["w"]
    batch = 1 if n <= 16 else n  # batch-1 like the reference on the small cases
    out = oracle.run_swarm(fit, pos, vel, int(g["t"]), w[0], w[1], w[2], tolerance=float(g["tolerance"]),
                           schedule_inertia=bool(g["schedule_inertia"]), early_stopping=bool(g["early_stopping"]),
                           batch=batch)
    iters = g["fitness"].shape[0]
    if batch != 1:
        # batched CPU convolutions round differently from batch-1 ones (~5e-6 on the fitness):
        # the trajectory is comparable while every decision has a larger margin than that
        safe = first_ambiguous_iteration(g["fitness"], 5e-5)
        assert safe >= (35 if name.endswith("margin") else 6)
        np.testing.assert_allclose(np.stack(out["fitness"])[:safe], g["fitness"][:safe], atol=TOL)
        np.testing.assert_allclose(np.stack(out["pos"])[:safe + 1], g["pos"][:safe + 1], rtol=1e-9, atol=1e-9)
        # and per step everywhere: the reference's positions in, the reference's fitness out
        for t in range(0, iters, 7):
            np.testing.assert_allclose(fit(g["pos"][t]), g["fitness"][t], atol=TOL)
        return
    assert out["i"] == int(g["i_ret"])
    assert len(out["fitness"]) == iters
    np.testing.assert_allclose(np.stack(out["fitness"]), g["fitness"], atol=TOL)
    np.testing.assert_allclose(np.stack(out["pos"]), g["pos"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(np.stack(out["vel"]), g["vel"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out["mean_mse"], g["mean_mse"], rtol=1e-12)
    np.testing.assert_allclose(np.array(out["g_best_val"], np.float32), g["g_best_val"], atol=TOL)
    np.testing.assert_allclose(np.array(out["g_best_val_dummy"], np.float32), g["g_best_val_dummy"], atol=TOL)
    np.testing.assert_allclose(out["g_best_pos"], g["g_best_pos"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out["pbest_pos"], g["pbest_pos"], rtol=1e-9, atol=1e-9)
    assert abs(out["w_inertia"] - float(g["w_inertia_final"])) < 1e-15


@pytest.mark.parametrize("name", ["inverter_in_n8_t5", "inverter_out_n8_t5"])
def test_inverter_variant(oracle, synth, name):
    """Encoder-seeded float32 positions, MSE-augmented fitness with eps counted twice and the
    float32 first move (reference util_pso.py:93-112, util_discovery.py:84-101)."""
    g = load_golden(name)
    gsd, csd = _weights(synth, g)
    zd, n = int(g["z_dim"]), int(g["n"])
    esd = synth.encoder_state(zd, 1, 64, seed=int(g["enc_seed"]))
    assert synth.checksum(esd) == str(g["enc_ck"])
    targets = torch.from_numpy(g["targets"])
    with torch.no_grad():
        lat = oracle.encoder_forward(synth.to_torch(esd), targets).numpy()
    np.testing.assert_allclose(lat, g["encoder_latents"], rtol=1e-4, atol=1e-4)
    # the swarm starts from the reference's own latents so that one encoder ulp cannot move the run
    pos0 = g["encoder_latents"].reshape(n, zd)
    np.testing.assert_array_equal(pos0.astype(np.float64), g["pos"][0])
    np.random.seed(int(g["np_seed"]))
    vel0 = oracle.init_velocities(n, zd)
    np.testing.assert_array_equal(vel0, g["vel"][0])
    fit = oracle.Fitness(gsd, csd, 1, str(g["mode"]), targets=targets)
    w = g["w"]
    out = oracle.run_swarm(fit, pos0, vel0, int(g["t"]), w[0], w[1], w[2], tolerance=float(g["tolerance"]),
                           schedule_inertia=False, early_stopping=False, batch=1, f32_first_move=True)
    np.testing.assert_allclose(np.stack(out["fitness"]), g["fitness"], atol=TOL)
    np.testing.assert_allclose(np.stack(out["pos"]), g["pos"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(np.stack(out["vel"]), g["vel"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(np.array(out["g_best_val_dummy"], np.float32), g["g_best_val_dummy"], atol=TOL)
    # the swarm arithmetic alone, fed the reference's fitness values and draws, is bit-exact
    calls = iter(g["fitness"])
    rep = oracle.run_swarm(lambda p, s: next(calls), g["pos"][0], g["vel"][0], int(g["t"]), w[0], w[1], w[2],
                           early_stopping=False, rand=g["rand"], batch=n, f32_first_move=True)
    np.testing.assert_array_equal(np.stack(rep["pos"]), g["pos"])
    np.testing.assert_array_equal(np.stack(rep["vel"]), g["vel"])


def test_rand_replay_matches_draw_order(oracle):
    """r1/r2 stored in the fixtures are the reference's own draws: feeding them back
    explicitly gives the same trajectory as drawing from the seeded global stream."""
    g = load_golden("swarm_n8_t5_z10")
    calls = iter(g["fitness"])
    out = oracle.run_swarm(lambda p, s: next(calls), g["pos"][0], g["vel"][0], int(g["t"]), *g["w"],
                           early_stopping=False, rand=g["rand"], batch=8)
    np.testing.assert_array_equal(np.stack(out["pos"]), g["pos"])


def test_toy_known_answer(oracle):
    """Reference hands_on/pso.py on the sphere function: converges to ~0 and stops early."""
    g = load_golden("toy_sphere")
    np.random.seed(42)
    r = np.stack([np.stack([np.random.uniform(-5, 5, 3), (np.random.randn(3) - 0.5) / 10]) for _ in range(20)])
    np.testing.assert_array_equal(r[:, 0], g["pos0"])
    sphere = lambda p, s: np.sum(p * p, axis=1)  # float64 objective, like the toy
    out = oracle.run_swarm(sphere, r[:, 0], r[:, 1], 100, *g["w"], tolerance=float(g["tolerance"]),
                           early_stopping=True, batch=20, diagnostics=False)
    np.testing.assert_allclose(np.array(out["g_best_val"], np.float64), g["g_best_val"], rtol=1e-6)
    np.testing.assert_allclose(out["g_best_pos"], g["g_best_pos"], rtol=1e-5, atol=1e-7)
    assert out["g_best_val"][-1] < 1e-3


def test_assessor_shapes(oracle, synth):
    g = load_golden("assessor_shapes")
    x = torch.from_numpy(g["x64"])
    for model in ("ResNet50", "ResNet101"):
        csd = synth.to_torch(synth.resnet_state(model, 1, 8, seed=12))
        with torch.no_grad():
            np.testing.assert_allclose(oracle.resnet_forward(csd, x).numpy(), g[f"{model}_logits64"], rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(oracle.resnet_forward(csd, x[:, :, :28, :28].contiguous()).numpy(),
                                       g[f"{model}_logits28"], rtol=1e-4, atol=1e-4)


def test_vq(oracle, synth):
    """Codebook seeding, nearest-code lookup and the VQ-VAE-GAN inference path against the
    reference's outputs (tests/golden/vq.npz)."""
    g = load_golden("vq")
    # seeding: float32 cast of the final positions, class blocks stacked
    cb = oracle.seed_codebook(list(g["A_final_positions"]))
    np.testing.assert_array_equal(cb, g["A_codebook"])
    wsd = synth.vqvae_gan_state(100, 64, 64, seed=int(g["A_seed"]))
    assert synth.checksum(wsd) == str(g["A_ck"])
    sd, cbt = synth.to_torch(wsd), torch.from_numpy(cb)
    with torch.no_grad():
        x_tilde, z_e_x, z_q_x, latents = oracle.vqvae_gan_forward(sd, cbt, torch.from_numpy(g["A_x"]))
        np.testing.assert_allclose(z_e_x.numpy(), g["A_z_e_x"], rtol=1e-4, atol=1e-4)
        np.testing.assert_array_equal(latents.numpy(), g["A_latents"])
        np.testing.assert_array_equal(z_q_x.numpy(), g["A_z_q_x"])
        np.testing.assert_allclose(x_tilde.numpy(), g["A_x_tilde"], atol=TOL)
        np.testing.assert_allclose(g["A_decode"], g["A_x_tilde"], atol=1e-6)  # decode(encode(x)) is the same image
        idx, _ = oracle.vq_nearest(torch.from_numpy(g["A_near"]), cbt)
        np.testing.assert_array_equal(idx.numpy(), g["A_near_idx"])
        # the fixtures' decisions are not near-ties: an fp32 reordering cannot flip them
        m = vq_margins(oracle.vq_distances(torch.from_numpy(g["A_z_e_x"]).reshape(-1, 100), cbt).numpy())
        assert m.min() > 1e-2, m.min()
        # the bare functions on a feature map, duplicated codes included (first index wins)
        cb2, inp = torch.from_numpy(g["B_codebook"]), torch.from_numpy(g["B_inputs"])
        idx, codes = oracle.vq_nearest(inp, cb2)
        np.testing.assert_array_equal(idx.numpy(), g["B_idx"])
        np.testing.assert_array_equal(codes.numpy(), g["B_codes"])
        np.testing.assert_array_equal(idx.reshape(-1).numpy(), g["B_idx_flat"])
        assert g["B_idx"][0, 0, 0] == 5 and g["B_idx"][1, 4, 4] == 0
        np.testing.assert_array_equal(g["B_emb_latents"], g["B_idx"])
        np.testing.assert_array_equal(g["B_emb_st"], np.transpose(g["B_codes"], (0, 3, 1, 2)))
        np.testing.assert_array_equal(g["B_emb_bar"], g["B_emb_st"])


def test_alexnet(oracle, synth):
    """The small "AlexNet" assessor (reference util_cnn.py:193-249): logits at 28x28 / 'same' / LeakyReLU
    and 64x64 / 'valid' / ReLU, fitness through the generator, and a short swarm, all from the
    reference's own classes."""
    g = load_golden("alexnet")
    sd28 = synth.to_torch(synth.alexnet_state(8, kernel=3, img_size=28, padding="same", seed=int(g["seed28"])))
    sd64 = synth.to_torch(synth.alexnet_state(8, kernel=3, img_size=64, padding="valid", seed=int(g["seed64"])))
    assert synth.checksum({k: v.numpy() for k, v in sd28.items()}) == str(g["ck28"])
    assert sd28["fc1.weight"].shape[1] == int(g["to_linear28"]) == 256 and sd64["fc1.weight"].shape[1] == int(g["to_linear64"]) == 1024
    with torch.no_grad():
        np.testing.assert_allclose(oracle.alexnet_forward(sd28, torch.from_numpy(g["x28"])).numpy(), g["logits28"], atol=1e-5)
        np.testing.assert_allclose(oracle.alexnet_forward(sd64, torch.from_numpy(g["x64"]), "valid", "ReLU").numpy(), g["logits64"], atol=1e-5)
    assert str(g["valid28_error"]) == "RuntimeError"
    gsd = synth.to_torch(synth.generator_state(10, 1, 64, seed=int(g["gen_seed"])))
    ci = sorted(g["iid"].tolist()).index(int(g["label"]))
    fit = oracle.Fitness(gsd, sd28, ci, cnn_args={"padding": "same", "cnn_activation": "LeakyReLU"})
    np.testing.assert_allclose(fit(g["z"]), g["fitness"], atol=1e-5)  # one batch of six here, six batches of one there
    ref = oracle.run_swarm(fit, g["sw_pos"][0], g["sw_vel"][0], int(g["sw_t"]), *[float(v) for v in g["sw_w"]], early_stopping=False,
                           rand=g["sw_rand"], batch=1)
    np.testing.assert_allclose(np.stack(ref["fitness"]), g["sw_fitness"], atol=1e-6)
    np.testing.assert_array_equal(np.stack(ref["pos"]), g["sw_pos"])
