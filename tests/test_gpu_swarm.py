"""GPU parity of the swarm loop against trajectories recorded from the reference
(tests/golden/swarm_*.npz, inverter_*.npz) -- through the reference-style classes, which
call the C ABI."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from parity_utils import first_ambiguous_iteration

pytestmark = pytest.mark.gpu
TOL = 1e-4  # fp32 outputs within 1e-4 of the reference CPU path
SWARMS = ["swarm_n8_t5_z10", "swarm_n64_t50_z10", "swarm_n16_t30_z100_sched_es", "swarm_n12_t40_z10_es",
          "swarm_n64_t50_z10_margin"]
# iterations a recorded run can be followed end to end (every decision clear of fp32 noise), at least
MIN_SAFE = {"swarm_n64_t50_z10_margin": 35}


@pytest.fixture(scope="module")
def mods():
    imp = lambda n: importlib.import_module("gan-discovery-pso_amd." + n)
    return dict(eng=imp("engine"), pso=imp("pso.util_pso"), disc=imp("pso.util_discovery"), cnn=imp("pso.util_cnn"),
                dcgan=imp("utils.util_dcgan"), inv=imp("inverter.utils_ae.util_inverter"), lib=imp("_lib"))


def build_discovery(mods, synth, g, mode="optimize_out_training"):
    dev = torch.device("cuda:0")
    zd, nc = int(g["z_dim"]), int(g["n_class"])
    gen = mods["dcgan"].Generator(zd, 1, 64).to(dev)
    gen.load_state_dict(synth.to_torch(synth.generator_state(zd, 1, 64, seed=int(g["gen_seed"]))))
    iid = g["iid"].tolist() if "iid" in g.files else [0, 1]
    cnn = mods["cnn"].get_cnn("ResNet50", 1, iid, nc).to(dev)
    cnn.load_state_dict(synth.to_torch(synth.resnet_state("ResNet50", 1, nc, seed=int(g["cnn_seed"]))))
    label = int(g["label"]) if "label" in g.files else 5
    return mods["disc"].Discovery(label, gen, cnn, dev, control_pso_fitness=mode,
                                  iid_classes=iid if "iid" in g.files else None, obj_fun_threshold=0.0)


@pytest.mark.parametrize("name", SWARMS)
def test_per_step_fitness(mods, synth, name):
    """The reference's positions in -> the reference's fitness out, at every recorded iteration."""
    g = load_golden(name)
    disc = build_discovery(mods, synth, g)
    for t in range(g["fitness"].shape[0]):
        got = disc.fitness_batch(g["pos"][t]).cpu().numpy()
        np.testing.assert_allclose(got, g["fitness"][t], atol=TOL, err_msg=f"iteration {t + 1}")


def test_single_particle_interface(mods, synth):
    """Discovery.fitness / particles_to_img keep the reference's return types and shapes."""
    g = load_golden("swarm_n8_t5_z10")
    disc = build_discovery(mods, synth, g)
    f, img01, img = disc.fitness(dim_space=10, pos=g["pos"][0][3])
    assert isinstance(f, np.ndarray) and f.dtype == np.float32 and f.shape == (1,)
    assert img01.shape == (1, 1, 28, 28) and img.shape == (1, 1, 28, 28)
    assert float(img01.min()) == 0.0 and float(img01.max()) == 1.0
    np.testing.assert_allclose(f, g["fitness"][0][3:4], atol=TOL)
    a, b = disc.particles_to_img(10, torch.from_numpy(g["pos"][0][3]))
    assert torch.equal(a, img01) and torch.equal(b, img)
    disc.control_pso_fitness = "something_else"
    with pytest.raises(ValueError):
        disc.fitness(dim_space=10, pos=g["pos"][0][3])


@pytest.mark.parametrize("name", SWARMS)
def test_trajectory(mods, synth, name):
    """Same seed in -> the reference's trajectory out, for as long as every strict '<'
    decision has a margin above fp32 rounding noise; the bookkeeping lists match too."""
    g = load_golden(name)
    disc = build_discovery(mods, synth, g)
    n, t, d = int(g["n"]), int(g["t"]), int(g["d"])
    w = g["w"]
    np.random.seed(int(g["np_seed"]))
    sw = mods["pso"].Swarm("/tmp/gdpso_test", disc, n, t, d, torch.device("cuda:0"), tolerance=float(g["tolerance"]),
                           w_inertia=w[0], w_cogn=w[1], w_soci=w[2])
    sw.verbose = False
    sw.record_fitness = True
    history, hp, hv, i_ret = sw.optimize(schedule_inertia=bool(g["schedule_inertia"]), early_stopping=bool(g["early_stopping"]))
    res = sw.results
    iters = g["fitness"].shape[0]
    safe = first_ambiguous_iteration(g["fitness"], 2e-5)
    assert safe >= min(iters, MIN_SAFE.get(name, 5))
    np.testing.assert_array_equal(res["hist_pos"][0], g["pos"][0])  # same RNG consumption
    np.testing.assert_array_equal(res["hist_vel"][0], g["vel"][0])
    np.testing.assert_allclose(res["fitness"][:safe], g["fitness"][:safe], atol=TOL)
    # float64 state, reference evaluation order, no FMA contraction: bit-identical
    np.testing.assert_array_equal(res["hist_pos"][:safe + 1], g["pos"][:safe + 1])
    np.testing.assert_array_equal(res["hist_vel"][:safe + 1], g["vel"][:safe + 1])
    np.testing.assert_allclose(res["mean_mse"][:safe], g["mean_mse"][:safe], rtol=1e-12)
    np.testing.assert_allclose(res["g_best_val_dummy"][:safe], g["g_best_val_dummy"][:safe], atol=TOL)
    if safe == iters:  # the whole run is comparable
        assert i_ret == int(g["i_ret"])
        assert len(history["mean_mse"]) == iters and history["global_best_val"] == []
        np.testing.assert_allclose(np.concatenate(sw.g_best_val), g["g_best_val"], atol=TOL)
        np.testing.assert_array_equal(sw.g_best_pos, g["g_best_pos"])
        np.testing.assert_array_equal(np.stack([p.p_best_pos for p in sw.swarm]), g["pbest_pos"])
        np.testing.assert_allclose(np.concatenate([p.p_best_val for p in sw.swarm]), g["pbest_val"], atol=TOL)
        assert abs(sw.w_inertia - float(g["w_inertia_final"])) < 1e-15
        assert sorted(hp.keys()) == sorted(g["hist_keys"].tolist())
        assert hp["particle_0"].shape == tuple(g["hist_shape"]) and hv["particle_0"].shape == tuple(g["hist_shape"])
        np.testing.assert_array_equal(hp[f"particle_{n - 1}"].to_numpy(), g["pos"][:, n - 1, :])
        # numpy's stream is left where the reference leaves it
        chk = np.random.random()
        np.random.seed(int(g["np_seed"]))
        np.random.randn(n, 2, d)
        np.random.random((iters, n, 2))
        assert chk == np.random.random()


@pytest.mark.parametrize("name", SWARMS + ["inverter_in_n8_t5", "inverter_out_n8_t5"])
def test_swarm_arithmetic_is_bit_exact(mods, name):
    """The swarm kernels alone: inject the reference's own fitness values and r1/r2 and
    compare every position, velocity, personal and global best bit for bit."""
    g = load_golden(name)
    eng, lib = mods["eng"], mods["lib"]
    n, d, iters = int(g["n"]), int(g["d"]), g["fitness"].shape[0]
    sw = object.__new__(eng.HipSwarm)
    ev = type("E", (), {"device": torch.device("cuda:0"), "gen": type("G", (), {"z_dim": d, "side": 28})()})()
    eng.HipSwarm.__init__(sw, ev, n, d, int(g["t"]) + 1)
    sw.f32_first_move = name.startswith("inverter")
    rand = np.concatenate([g["rand"], g["rand"][-1:]])  # one spare row for the extra iteration below
    sw.load(g["pos"][0], g["vel"][0], float(g["w"][0]), rand=rand)
    for i in range(1, iters + 1):
        sw.fitness.copy_(torch.from_numpy(g["fitness"][i - 1]))
        sw.local_best(i)
        sw.global_best(i, bool(g["schedule_inertia"]), bool(g["early_stopping"]), float(g["tolerance"]))
        sw.move(i, float(g["w"][1]), float(g["w"][2]))
        sw.mean_distance(i, sw.pos)
    # one more iteration than the reference ran: must be a no-op when it stopped early
    if iters < int(g["t"]):
        sw.fitness.zero_()
        sw.local_best(iters + 1)
        sw.global_best(iters + 1, False, True, float(g["tolerance"]))
        sw.move(iters + 1, float(g["w"][1]), float(g["w"][2]))
    res = sw.results()
    assert res["iters"] == iters
    assert res["stopped"] == (int(g["i_ret"]) <= int(g["t"]))
    np.testing.assert_array_equal(res["hist_pos"], g["pos"])
    np.testing.assert_array_equal(res["hist_vel"], g["vel"])
    np.testing.assert_array_equal(res["g_best_val"], g["g_best_val"])
    np.testing.assert_array_equal(res["g_best_val_dummy"], g["g_best_val_dummy"])
    np.testing.assert_array_equal(res["g_best_pos"], g["g_best_pos"])
    np.testing.assert_array_equal(res["pbest_val"], g["pbest_val"])
    np.testing.assert_array_equal(res["pbest_pos"], g["pbest_pos"])
    np.testing.assert_allclose(res["mean_mse"], g["mean_mse"], rtol=1e-13)
    assert res["w_inertia"] == float(g["w_inertia_final"])


@pytest.mark.parametrize("name,mode", [("inverter_in_n8_t5", "optimize_in_training"), ("inverter_out_n8_t5", "optimize_out_training")])
def test_inverter_variant(mods, synth, name, mode):
    """Encoder-seeded positions + MSE-augmented fitness (reference pso_inverter.py:305-321)."""
    g = load_golden(name)
    dev = torch.device("cuda:0")
    disc = build_discovery(mods, synth, g, mode=mode)
    enc = mods["inv"].Encoder(int(g["z_dim"]), 1).to(dev)
    enc.load_state_dict(synth.to_torch(synth.encoder_state(int(g["z_dim"]), 1, 64, seed=int(g["enc_seed"]))))
    targets = torch.from_numpy(g["targets"])  # [n,1,28,28]
    lat = enc(targets.to(dev)).cpu().numpy()
    np.testing.assert_allclose(lat, g["encoder_latents"], rtol=1e-4, atol=1e-4)
    # per step, with the reference's positions and targets
    for t in range(g["fitness"].shape[0]):
        got = disc.fitness_batch(g["pos"][t], target=targets[:, 0].to(dev)).cpu().numpy()
        np.testing.assert_allclose(got, g["fitness"][t], atol=TOL)
    f1, _, rec = disc.fitness_pso_inverter(10, g["pos"][1][2], targets[2].numpy())
    assert f1.shape == (1,) and f1.dtype == np.float32 and rec.shape == (1, 1, 28, 28)
    np.testing.assert_allclose(f1, g["fitness"][1][2:3], atol=TOL)
    # whole run through Swarm.optimize(ood_patient=..., encoder=...)
    n, t, d = int(g["n"]), int(g["t"]), int(g["d"])
    loader = [(targets[k:k + 1], 0) for k in range(n)]
    np.random.seed(int(g["np_seed"]))
    sw = mods["pso"].Swarm("/tmp/gdpso_test", disc, n, t, d, dev, tolerance=float(g["tolerance"]), w_inertia=g["w"][0],
                           w_cogn=g["w"][1], w_soci=g["w"][2])
    sw.verbose = False
    sw.record_fitness = True
    _, hp, _, i_ret = sw.optimize(schedule_inertia=False, early_stopping=False, ood_patient=loader, encoder=enc)
    res = sw.results
    assert i_ret == int(g["i_ret"])
    np.testing.assert_allclose(res["fitness"], g["fitness"], atol=TOL)
    # initial positions are the encoder's float32 outputs: close, not bit-equal, so the
    # trajectory is compared to float32 resolution
    np.testing.assert_allclose(res["hist_pos"], g["pos"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(res["hist_vel"][0], g["vel"][0])
    np.testing.assert_allclose(np.concatenate(sw.g_best_val), g["g_best_val"], atol=TOL)


@pytest.mark.parametrize("n_shards,table", [(3, False), (2, False), (2, True), (4, True)])
def test_logical_shards_match_unsharded(mods, n_shards, table):
    """Shards of one swarm on one device, exchanging their tables the way the in-place all-gather does
    (the phases in the order of engine.sharded_iterations: positions gathered behind every move and the
    diagnostic one exchange behind, or positions in the exchange table and the diagnostic two behind),
    take exactly the decisions of the unsharded swarm."""
    g = load_golden("swarm_n64_t50_z10")
    eng = mods["eng"]
    n, d, iters = int(g["n"]), int(g["d"]), g["fitness"].shape[0]
    ev = type("E", (), {"device": torch.device("cuda:0"), "gen": type("G", (), {"z_dim": d, "side": 28})()})()
    bounds = [eng.shard_bounds(n, n_shards, r) for r in range(n_shards)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n
    shards = []
    for r, (lo, hi) in enumerate(bounds):
        s = object.__new__(eng.HipSwarm)
        eng.HipSwarm.__init__(s, ev, hi - lo, d, int(g["t"]), first_index=lo, n_ranks=n_shards, rank=r, n_global=n, table_positions=table)
        s.load(g["pos"][0][lo:hi], g["vel"][0][lo:hi], float(g["w"][0]), rand=g["rand"][:, lo:hi])
        shards.append(s)
    wc, ws_ = float(g["w"][1]), float(g["w"][2])

    def exchange():  # every rank's own row to everybody
        rows = [s.exchange_table[s.rank].clone() for s in shards]
        for s in shards:
            for r, row in enumerate(rows):
                if r != s.rank:
                    s.exchange_table[r].copy_(row)

    def fitness(i):
        for s, (lo, hi) in zip(shards, bounds):
            s.fitness.copy_(torch.from_numpy(g["fitness"][i - 1][lo:hi]))
            s.local_best(i)

    if table:
        for i in range(1, iters + 1):
            fitness(i)
            exchange()
            for s in shards:
                if i > 1:
                    s.distance_part_table(i - 1)
                s.after_exchange(i, False, False, 1e-5, wc, ws_, diag_iteration=max(0, i - 2))
        exchange()
        for s in shards:
            s.distance_finish(iters - 1)
            s.distance_part_table(iters)
        exchange()
        for s in shards:
            s.distance_finish(iters)
    else:
        allpos = None
        for i in range(1, iters + 1):
            fitness(i)
            if i > 1:
                for s in shards:
                    s.distance_part(i - 1, allpos)
            exchange()
            for s in shards:
                s.after_exchange(i, False, False, 1e-5, wc, ws_, diag_iteration=i - 1)
            allpos = torch.cat([s.pos for s in shards])
        for s in shards:
            s.distance_part(iters, allpos)
        exchange()
        for s in shards:
            s.distance_finish(iters)
    res = [s.results() for s in shards]
    np.testing.assert_array_equal(np.concatenate([r["hist_pos"] for r in res], axis=1), g["pos"])
    for r in res:
        np.testing.assert_array_equal(r["g_best_val"], g["g_best_val"])
        np.testing.assert_array_equal(r["g_best_pos"], g["g_best_pos"])
        np.testing.assert_allclose(r["mean_mse"], g["mean_mse"], rtol=1e-13)
        np.testing.assert_array_equal(r["mean_mse"], res[0]["mean_mse"])  # every rank, the same bits


@pytest.mark.parametrize("n,d,parts", [(2, 10, 2), (17, 33, 3), (333, 100, 3), (1000, 100, 2), (2100, 7, 8), (4096, 100, 8)])
def test_distance_diagnostic_sizes(mods, n, d, parts):
    """Swarm.mse at every tile side (16, 32, 64), with coordinate counts that are and are not multiples
    of the chunk, tiles past the last particle, and the sizes a weak-scaling run reaches: the whole
    diagnostic on one rank and the same cut into `parts` shares against float64 on the host."""
    eng = mods["eng"]
    ev = type("E", (), {"device": torch.device("cuda:0"), "gen": type("G", (), {"z_dim": d, "side": 28})()})()
    rs = np.random.RandomState(n + d)
    pos = torch.from_numpy(rs.standard_normal((n, d)) * rs.uniform(0.1, 3.0, (n, 1))).cuda()
    want = torch.pdist(pos.cpu()).mean().item()
    local = max(1, n // parts)
    swarms = []
    for r in range(parts):
        s = object.__new__(eng.HipSwarm)
        eng.HipSwarm.__init__(s, ev, local, d, 2, first_index=0, n_ranks=parts, rank=r, n_global=n, history=False)
        s.load(np.zeros((local, d)), np.zeros((local, d)), 0.7)
        s.distance_part(1, pos)
        s.mean_distance(2, pos)  # the whole diagnostic on this rank
        swarms.append(s)
    table = torch.stack([s.cand[s.rank] for s in swarms])
    for s in swarms:
        s.cand.copy_(table)
        s.distance_finish(1)
    if n % parts == 0:  # the same once more with the positions read from the exchange table
        tabled = []
        for r in range(parts):
            s = object.__new__(eng.HipSwarm)
            eng.HipSwarm.__init__(s, ev, n // parts, d, 2, first_index=0, n_ranks=parts, rank=r, n_global=n, history=False,
                                  table_positions=True)
            s.load(np.zeros((n // parts, d)), np.zeros((n // parts, d)), 0.7)
            cd = s.cand.shape[1]
            s.exchange_table[:, cd:].copy_(pos.view(parts, -1))
            s.distance_part_table(1)
            tabled.append(s)
        for r, s in enumerate(tabled):
            assert s.cand[r, d + 3].item() == swarms[r].cand[r, d + 3].item()  # iteration 1: the odd slot
    got = torch.stack([s.mean_dist for s in swarms]).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-13)
    assert (got[:, 0] == got[0, 0]).all() and (got[:, 1] == got[0, 1]).all()  # every rank, the same bits


@pytest.mark.parametrize("n", [40, 300])
def test_diagnostic_of_the_fused_loop_in_any_chunking(mods, synth, n):
    """The fused loop adds up an iteration's tile sums in the NEXT iteration's distance launch (the last one's in a
    launch of its own): whole, in chunks of one and in uneven chunks it leaves the same bits, and they are the mean
    distance of the positions the history holds; with early stopping nothing is written past the stop."""
    eng = mods["eng"]
    dev = torch.device("cuda:0")
    d, iters = 10, 9
    ev = eng.Evaluator(eng.PackedGenerator(synth.generator_state(d, 1, 64, seed=11), dev),
                       eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=28))
    spec = eng.FitnessSpec(class_index=2)
    r = np.random.RandomState(3).standard_normal((n, 2, d))

    def run(chunks, early=False, tol=1e-5):
        sw = eng.HipSwarm(ev, n, d, iters, history=True, diagnostics=True)
        sw.load(r[:, 0], (r[:, 1] - 0.5) / 10, 0.73)
        first = 1
        for count in chunks:
            sw.iterate(first, count, spec, 1.496, 1.496, True, early, tol, True, seed=5)
            first += count
        torch.cuda.synchronize()
        res = sw.results()
        return res

    whole = run([iters])
    for chunks in ([1] * iters, [2, 1, 4, 2], [8, 1]):
        got = run(chunks)
        np.testing.assert_array_equal(got["mean_mse"], whole["mean_mse"])
        np.testing.assert_array_equal(got["hist_pos"], whole["hist_pos"])
    hp = np.asarray(whole["hist_pos"])  # [iters + 1, n, d]
    want = [torch.pdist(torch.from_numpy(np.ascontiguousarray(hp[i]))).mean().item() for i in range(1, iters + 1)]
    np.testing.assert_allclose(np.asarray(whole["mean_mse"])[:iters], want, rtol=1e-13)
    # a tolerance no improvement reaches: the loop stops early, in every chunking at the same iteration with the same values
    stopped = run([iters], early=True, tol=1e9)
    again = run([1] * iters, early=True, tol=1e9)
    assert stopped["stopped"] and 1 <= stopped["iters"] < iters and again["iters"] == stopped["iters"]
    np.testing.assert_array_equal(again["mean_mse"], stopped["mean_mse"])
    np.testing.assert_array_equal(stopped["mean_mse"], np.asarray(whole["mean_mse"])[:stopped["iters"]])


def test_many_particles_take_the_multi_workgroup_kernels(mods):
    """More particles than the one-workgroup kernels take (4096): personal bests, candidate record,
    global best and move through pbest_kernel / candidate_kernel / global_best_kernel / move_kernel,
    against numpy float64 on the same fitness values and r1/r2."""
    eng = mods["eng"]
    n, d, iters = 5000, 10, 4
    rs = np.random.RandomState(17)
    pos, vel = rs.standard_normal((n, d)), rs.standard_normal((n, d)) * 0.1
    fit = rs.random_sample((iters, n)).astype(np.float32)
    fit[1, 100] = fit[1, 4000] = np.float32(-1.0)  # a tie between two workgroups: the lower index wins
    rand = rs.random_sample((iters, n, 2))
    w, wc, ws_ = 0.73, 1.496, 1.3
    ev = type("E", (), {"device": torch.device("cuda:0"), "gen": type("G", (), {"z_dim": d, "side": 28})()})()
    sw = object.__new__(eng.HipSwarm)
    eng.HipSwarm.__init__(sw, ev, n, d, iters, diagnostics=False)
    sw.load(pos, vel, w, rand=rand)
    pb_val, pb_pos = np.full(n, np.inf, np.float32), pos.copy()
    g_val, g_pos = np.float32(np.inf), np.zeros(d)
    for i in range(1, iters + 1):
        sw.fitness.copy_(torch.from_numpy(fit[i - 1]))
        sw.local_best(i)
        sw.global_best(i, False, False, 1e-5)
        sw.move(i, wc, ws_)
        better = fit[i - 1] < pb_val
        pb_val = np.where(better, fit[i - 1], pb_val)
        pb_pos = np.where(better[:, None], pos, pb_pos)
        k = int(np.argmin(pb_val))
        if pb_val[k] < g_val:
            g_val, g_pos = pb_val[k], pb_pos[k].copy()
        r1, r2 = rand[i - 1][:, 0:1], rand[i - 1][:, 1:2]
        vel = (w * vel + (ws_ * r2) * (pb_pos - pos)) + (wc * r1) * (g_pos[None] - pos)
        pos = pos + vel
        cand = sw.cand.cpu().numpy()[0]
        assert cand[0] == pb_val[k] and int(cand[1]) == k
    res = sw.results()
    assert res["gbest_val"] == g_val and res["gbest_idx"] == 100
    np.testing.assert_array_equal(res["pbest_val"], pb_val)
    np.testing.assert_array_equal(res["pbest_pos"], pb_pos)
    np.testing.assert_array_equal(res["g_best_pos"], g_pos)
    np.testing.assert_array_equal(res["pos"], pos)
    np.testing.assert_array_equal(res["vel"], vel)


@pytest.mark.parametrize("table", [False, True])
def test_sharded_large_swarm(mods, table):
    """Two logical ranks of 2600 particles each (too many for the one-workgroup kernels: after_exchange
    falls back to distance_finish + global_best + move, the distance tiles are 64 x 64 and, with the
    positions in the exchange table, cross the boundary between the ranks' blocks) against numpy
    float64 on the same fitness values and r1/r2."""
    eng = mods["eng"]
    n_loc, ranks, d, iters = 2600, 2, 10, 4
    n = n_loc * ranks
    rs = np.random.RandomState(23)
    pos, vel = rs.standard_normal((n, d)), rs.standard_normal((n, d)) * 0.1
    fit = rs.random_sample((iters, n)).astype(np.float32)
    fit[2, 17] = fit[2, 3000] = np.float32(-1.0)  # a tie between the two ranks: the lower index wins
    rand = rs.random_sample((iters, n, 2))
    w, wc, ws_ = 0.73, 1.496, 1.3
    ev = type("E", (), {"device": torch.device("cuda:0"), "gen": type("G", (), {"z_dim": d, "side": 28})()})()
    shards = []
    for r in range(ranks):
        s = object.__new__(eng.HipSwarm)
        eng.HipSwarm.__init__(s, ev, n_loc, d, iters, first_index=r * n_loc, n_ranks=ranks, rank=r, n_global=n, history=False,
                              table_positions=table)
        s.load(pos[r * n_loc:(r + 1) * n_loc], vel[r * n_loc:(r + 1) * n_loc], w, rand=rand[:, r * n_loc:(r + 1) * n_loc])
        shards.append(s)

    def exchange():
        rows = [s.exchange_table[s.rank].clone() for s in shards]
        for s in shards:
            for r, row in enumerate(rows):
                if r != s.rank:
                    s.exchange_table[r].copy_(row)

    pb_val, pb_pos = np.full(n, np.inf, np.float32), pos.copy()
    g_val, g_pos = np.float32(np.inf), np.zeros(d)
    want_dist, allpos = [], None
    for i in range(1, iters + 1):
        for r, s in enumerate(shards):
            s.fitness.copy_(torch.from_numpy(fit[i - 1][r * n_loc:(r + 1) * n_loc]))
            s.local_best(i)
            if not table and i > 1:
                s.distance_part(i - 1, allpos)
        exchange()
        for s in shards:
            if table and i > 1:
                s.distance_part_table(i - 1)
            s.after_exchange(i, False, False, 1e-5, wc, ws_, diag_iteration=max(0, i - 2) if table else i - 1)
        allpos = torch.cat([s.pos for s in shards])
        better = fit[i - 1] < pb_val
        pb_val = np.where(better, fit[i - 1], pb_val)
        pb_pos = np.where(better[:, None], pos, pb_pos)
        k = int(np.argmin(pb_val))
        if pb_val[k] < g_val:
            g_val, g_pos = pb_val[k], pb_pos[k].copy()
        r1, r2 = rand[i - 1][:, 0:1], rand[i - 1][:, 1:2]
        vel = (w * vel + (ws_ * r2) * (pb_pos - pos)) + (wc * r1) * (g_pos[None] - pos)
        pos = pos + vel
        want_dist.append(torch.pdist(torch.from_numpy(pos)).mean().item())
    if table:
        exchange()
        for s in shards:
            s.distance_finish(iters - 1)
            s.distance_part_table(iters)
    else:
        for s in shards:
            s.distance_part(iters, allpos)
    exchange()
    for s in shards:
        s.distance_finish(iters)
    res = [s.results() for s in shards]
    np.testing.assert_array_equal(np.concatenate([r["pos"] for r in res]), pos)
    np.testing.assert_array_equal(np.concatenate([r["vel"] for r in res]), vel)
    for r in res:
        assert r["gbest_val"] == g_val and r["gbest_idx"] == 17
        np.testing.assert_array_equal(r["g_best_pos"], g_pos)
        np.testing.assert_allclose(r["mean_mse"], want_dist, rtol=1e-13)
        np.testing.assert_array_equal(r["mean_mse"], res[0]["mean_mse"])


@pytest.mark.parametrize("beside,table", [(False, False), (True, False), (False, True)])
def test_sharded_loop_over_rccl(mods, beside, table):
    """engine.sharded_iterations through a real RCCL process group (one rank: all this box has), with
    the positions' all-gather on the launch stream (small swarms), on the backend's own stream beside
    the next iteration (large ones), or the positions in the exchange table (no gather of their own):
    the position all-gather started behind every move, the distance sums in the next candidate
    records, the deferred distance diagnostic and the early stop reproduce the reference trajectory bit for bit."""
    import socket
    import torch.distributed as dist
    eng = mods["eng"]
    g = load_golden("swarm_n64_t50_z10")
    n, d, iters, t = int(g["n"]), int(g["d"]), g["fitness"].shape[0], int(g["t"])
    ev = type("E", (), {"device": torch.device("cuda:0"), "gen": type("G", (), {"z_dim": d, "side": 28})()})()

    class Injected(eng.HipSwarm):
        def evaluate(self, i, spec):  # the reference's own fitness values instead of the networks
            self.fitness.copy_(torch.from_numpy(g["fitness"][min(i, iters) - 1]))

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        comm = eng.TorchComm()
        assert not comm.host_staged
        comm.async_bytes = 0 if beside else 1 << 40
        sw = Injected(ev, n, d, t + 1, first_index=0, n_ranks=1, rank=0, n_global=n, table_positions=table)
        rand = np.concatenate([g["rand"], np.repeat(g["rand"][-1:], t + 1 - iters, axis=0)])
        sw.load(g["pos"][0], g["vel"][0], float(g["w"][0]), rand=rand)
        # two calls, as bench.py makes them (warm-up, then the timed steps): the diagnostic of the
        # first call's last iteration is flushed at its end
        eng.sharded_iterations(sw, comm, 1, 3, None, float(g["w"][1]), float(g["w"][2]), bool(g["schedule_inertia"]),
                               bool(g["early_stopping"]), float(g["tolerance"]), True, poll_every=4)
        eng.sharded_iterations(sw, comm, 4, t - 3, None, float(g["w"][1]), float(g["w"][2]), bool(g["schedule_inertia"]),
                               bool(g["early_stopping"]), float(g["tolerance"]), True, poll_every=4)
        res = sw.results()
    finally:
        dist.destroy_process_group()
    assert res["iters"] == iters
    np.testing.assert_array_equal(res["hist_pos"], g["pos"])
    np.testing.assert_array_equal(res["g_best_val"], g["g_best_val"])
    np.testing.assert_array_equal(res["g_best_pos"], g["g_best_pos"])
    np.testing.assert_allclose(res["mean_mse"], g["mean_mse"], rtol=1e-13)


def test_full_size_properties(mods, synth):
    """512 particles x 10 iterations (the benchmark shape) with device-side r1/r2:
    size-independent invariants of the loop."""
    g = load_golden("swarm_n8_t5_z10")
    disc = build_discovery(mods, synth, g)
    n, t, d = 512, 10, 10
    np.random.seed(1)
    sw = mods["pso"].Swarm("/tmp/gdpso_test", disc, n, t, d, torch.device("cuda:0"), w_inertia=0.73, w_cogn=1.496, w_soci=1.496)
    sw.verbose = False
    sw.record_fitness = True
    sw.device_rng_seed = 1234
    _, hp, hv, i_ret = sw.optimize(schedule_inertia=False, early_stopping=False)
    res = sw.results
    assert i_ret == t + 1 and len(hp) == n and hp["particle_511"].shape == (t + 1, d)
    fit = res["fitness"]
    assert fit.shape == (t, n) and np.all(fit >= 0.1 - 1e-6) and np.all(fit <= 1.1 + 1e-6)
    np.testing.assert_array_equal(res["pbest_val"], fit.min(axis=0))            # personal best = min over own history
    np.testing.assert_array_equal(res["g_best_val_dummy"], np.minimum.accumulate(fit.min(axis=1)))
    assert np.all(np.diff(res["g_best_val"]) < 0)                               # grows on strict improvement only
    k = int(np.argmin(res["pbest_val"]))
    assert res["gbest_idx"] == k                                                # lowest index among the minima
    np.testing.assert_array_equal(res["g_best_pos"], res["pbest_pos"][k])
    t_best = fit[:, k].argmin()
    np.testing.assert_array_equal(res["pbest_pos"][k], res["hist_pos"][t_best, k])
    np.testing.assert_array_equal(res["hist_pos"][1:], res["hist_pos"][:-1] + res["hist_vel"][1:])  # pos += vel
    # re-running with the same device seed reproduces the run bit for bit
    np.random.seed(1)
    sw2 = mods["pso"].Swarm("/tmp/gdpso_test", disc, n, t, d, torch.device("cuda:0"), w_inertia=0.73, w_cogn=1.496, w_soci=1.496)
    sw2.verbose = False
    sw2.device_rng_seed = 1234
    sw2.optimize(schedule_inertia=False, early_stopping=False)
    np.testing.assert_array_equal(sw2.results["hist_pos"], res["hist_pos"])


def test_device_rng_is_uniform_and_shard_independent(mods):
    eng = mods["eng"]
    d, n = 4, 4096
    ev = type("E", (), {"device": torch.device("cuda:0"), "gen": type("G", (), {"z_dim": d, "side": 28})()})()

    def one_move(lo, hi):
        s = object.__new__(eng.HipSwarm)
        eng.HipSwarm.__init__(s, ev, hi - lo, d, 2, first_index=lo, n_ranks=1, rank=0, n_global=n)
        s.load(np.zeros((hi - lo, d)), np.zeros((hi - lo, d)), 0.0)
        s.pbest_pos.fill_(1.0)        # pos = 0, gbest = 0, pbest = 1  ->  new vel = w_soci * r2
        s.move(1, 0.0, 1.0, seed=99)
        a = s.vel.cpu().numpy().copy()
        s.load(np.zeros((hi - lo, d)), np.zeros((hi - lo, d)), 0.0)
        s.gbest_pos.fill_(1.0)        # -> new vel = w_cogn * r1
        s.move(1, 1.0, 0.0, seed=99)
        return a, s.vel.cpu().numpy().copy()

    r2, r1 = one_move(0, n)
    for r in (r1, r2):
        assert np.all(r >= 0) and np.all(r < 1)
        assert np.all(r == r[:, :1])                       # one scalar per particle, not per dimension
        assert abs(r[:, 0].mean() - 0.5) < 0.02 and abs(r[:, 0].var() - 1 / 12) < 0.01
    assert abs(np.corrcoef(r1[:, 0], r2[:, 0])[0, 1]) < 0.05
    r2b, r1b = one_move(1000, 1100)
    np.testing.assert_array_equal(r1b, r1[1000:1100])      # keyed by global particle index
    np.testing.assert_array_equal(r2b, r2[1000:1100])


def test_no_cpu_fallback(mods):
    with pytest.raises(mods["lib"].GdpsoError):
        mods["eng"].PackedGenerator({}, torch.device("cpu"))
