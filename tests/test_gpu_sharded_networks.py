"""The sharded loop THROUGH THE NETWORKS: every rank's HipSwarm.evaluate runs the generator and the
assessor on its own block of particles (nothing injected), the ranks exchange their candidate records,
and the run is compared with the unsharded one and with the reference's trajectories.

Logical ranks (tests/logical_ranks.py) stand in for GPUs: one thread per rank, all on this one device.
Reference: the loop of src/pso/util_pso.py:114-151,167-193, whose only coupling between particles is
the global best.

What "the same result whatever the number of ranks" means here: the swarm arithmetic is float64 and
order-fixed, the draws are keyed by global particle index, so the only thing that can differ is the
last bits of an fp32 fitness value, and that only because the convolution planner picks tile shapes
by batch size (a 32-row tile adds its K parts in another order than a 64-row one).  With the plan
pinned to one batch size (engine.plan_options) the sharded and the unsharded run are bit-identical;
with every batch planned by its own size they agree until the first decision closer than fp32 noise.
"""
import copy
import importlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from logical_ranks import run_ranks
from parity_utils import first_ambiguous_iteration

pytestmark = pytest.mark.gpu
TOL = 1e-4
W = (0.73, 1.496, 1.496)


@pytest.fixture(scope="module")
def mods():
    imp = lambda n: importlib.import_module("gan-discovery-pso_amd." + n)
    return dict(eng=imp("engine"), pso=imp("pso.util_pso"), disc=imp("pso.util_discovery"), cnn=imp("pso.util_cnn"),
                dcgan=imp("utils.util_dcgan"), inv=imp("inverter.utils_ae.util_inverter"), lib=imp("_lib"))


def _networks(eng, synth, z_dim, side, dtype):
    dev = torch.device("cuda:0")
    gen = eng.PackedGenerator(synth.generator_state(z_dim, 1, 64, seed=11, first_kernel=side // 4), dev, dtype=dtype)
    ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=side, dtype=dtype)
    return gen, ass


def _run_sharded(eng, gen, ass, world, p, d, iters, pos0, vel0, spec, table):
    n = world * p

    def rank_fn(rank, comm):
        ev = eng.Evaluator(gen, ass)  # buffers of its own; the packed weights are shared
        lo = rank * p
        sw = eng.HipSwarm(ev, p, d, iters, first_index=lo, n_ranks=world, rank=rank, n_global=n, history=True, diagnostics=True,
                          record_fitness=True, table_positions=table)
        sw.load(pos0[lo:lo + p], vel0[lo:lo + p], W[0])
        eng.run_loop(sw, iters, spec, W[1], W[2], schedule_inertia=False, early_stopping=False, tolerance=1e-5, diagnostics=True,
                     seed=42, comm=comm)
        res = sw.results()
        res["exchanges"] = comm.exchanges
        return res

    return run_ranks(world, rank_fn)


def _run_unsharded(eng, gen, ass, n, d, iters, pos0, vel0, spec):
    ev = eng.Evaluator(gen, ass)
    sw = eng.HipSwarm(ev, n, d, iters, history=True, diagnostics=True, record_fitness=True)
    sw.load(pos0, vel0, W[0])
    eng.run_loop(sw, iters, spec, W[1], W[2], schedule_inertia=False, early_stopping=False, tolerance=1e-5, diagnostics=True, seed=42)
    return sw.results()


def _assert_same_run(ranks, whole, p, exact_fitness=True, upto=None):
    """Every rank's view against the unsharded run: bit for bit (up to iteration `upto`, all when None)."""
    fit = np.concatenate([r["fitness"] for r in ranks], axis=1)
    hist = np.concatenate([r["hist_pos"] for r in ranks], axis=1)
    t = fit.shape[0] if upto is None else upto
    if exact_fitness:
        np.testing.assert_array_equal(fit[:t], whole["fitness"][:t])
    else:
        np.testing.assert_allclose(fit[:t], whole["fitness"][:t], atol=TOL)
    np.testing.assert_array_equal(hist[:t + 1], whole["hist_pos"][:t + 1])
    np.testing.assert_array_equal(np.concatenate([r["hist_vel"] for r in ranks], axis=1)[:t + 1], whole["hist_vel"][:t + 1])
    for r in ranks:
        if exact_fitness:
            np.testing.assert_array_equal(r["g_best_val_dummy"][:t], whole["g_best_val_dummy"][:t])
        else:
            np.testing.assert_allclose(r["g_best_val_dummy"][:t], whole["g_best_val_dummy"][:t], atol=TOL)
        np.testing.assert_allclose(r["mean_mse"][:t], whole["mean_mse"][:t], rtol=1e-13)
        np.testing.assert_array_equal(r["mean_mse"], ranks[0]["mean_mse"])  # every rank, the same bits
        if upto is None:
            assert r["gbest_idx"] == whole["gbest_idx"] and r["gbest_val"] == whole["gbest_val"]
            np.testing.assert_array_equal(r["g_best_val"], whole["g_best_val"])
            np.testing.assert_array_equal(r["g_best_pos"], whole["g_best_pos"])
    if upto is None:
        np.testing.assert_array_equal(np.concatenate([r["pbest_val"] for r in ranks]), whole["pbest_val"])
        np.testing.assert_array_equal(np.concatenate([r["pbest_pos"] for r in ranks]), whole["pbest_pos"])


@pytest.mark.parametrize("world,p,z_dim,side,dtype,iters,table", [
    (8, 512, 10, 28, "f32", 5, True),     # 4096 particles over 8 ranks, fp32, 28x28
    (8, 256, 10, 64, "f16", 3, True),     # 2048 particles over 8 ranks, fp16, 64x64
    (4, 96, 100, 28, "bf16", 4, False),   # positions in a gather of their own; Z = 100, bf16
    (3, 768, 10, 28, "f32", 3, True),     # one and a half rounds of the resident launches' workgroups per rank (a short last
                                          # round with one image a workgroup), four and a half on the single rank
])
def test_sharded_run_equals_unsharded_run(mods, synth, world, p, z_dim, side, dtype, iters, table):
    """8 logical ranks, every one evaluating its own block through the real generator and assessor, against the
    whole swarm on one rank.  Planned for the ranks' block size on both sides: bit for bit, every particle's
    whole history, every fitness value, the global-best sequence.  With the unsharded run planned for its own
    (larger) batch the fitness values agree to fp32 rounding and the runs agree up to the first near-tie."""
    eng = mods["eng"]
    n = world * p
    rs = np.random.RandomState(7)
    r = rs.standard_normal((n, 2, z_dim))
    pos0, vel0 = r[:, 0].copy(), (r[:, 1] - 0.5) / 10
    spec = eng.FitnessSpec(class_index=6)
    eng.plan_options(plan_images=p)
    try:
        gen, ass = _networks(eng, synth, z_dim, side, dtype)
        ranks = _run_sharded(eng, gen, ass, world, p, z_dim, iters, pos0, vel0, spec, table)
        pinned = _run_unsharded(eng, gen, ass, n, z_dim, iters, pos0, vel0, spec)
    finally:
        eng.plan_options(plan_images=0)
    # one exchange per iteration; the deferred diagnostic adds two at the end (one with a position gather of its own)
    assert all(r["exchanges"] == iters + (2 if table else 1) for r in ranks)
    assert all(r["iters"] == iters for r in ranks)
    _assert_same_run(ranks, pinned, p)
    # every batch planned by its own size: the same run to fp32 rounding
    gen, ass = _networks(eng, synth, z_dim, side, dtype)
    natural = _run_unsharded(eng, gen, ass, n, z_dim, iters, pos0, vel0, spec)
    # 16-bit storage: another summation order rounds the stored activations differently; the bars are those of
    # tests/test_gpu_lowprec.py against the fp32 reference (a quantity in [0.1, 1.1])
    bar = {"f32": TOL, "f16": 0.025, "bf16": 0.12}[dtype]
    np.testing.assert_allclose(natural["fitness"][0], pinned["fitness"][0], atol=bar)  # iteration 1: the same positions in
    if dtype == "f32":
        safe = first_ambiguous_iteration(pinned["fitness"], 2e-5)
        assert safe >= 1
        _assert_same_run(ranks, natural, p, exact_fitness=False, upto=safe)


def test_swarm_optimize_on_eight_ranks_follows_the_reference(mods, synth):
    """Swarm.optimize itself, sharded 8 x 8 (its ``comm`` knob), on the recorded 64-particle run whose decisions
    stay clear of fp32 noise for 35 iterations: every rank returns the reference's trajectory."""
    from test_gpu_swarm import build_discovery
    g = load_golden("swarm_n64_t50_z10_margin")
    n, t, d = int(g["n"]), int(g["t"]), int(g["d"])
    base = build_discovery(mods, synth, g)
    ev0 = base.evaluator()
    world = 8

    def rank_fn(rank, comm):
        disc = copy.copy(base)
        disc._evaluator = mods["eng"].Evaluator(ev0.gen, ev0.ass)  # same packed weights, buffers of its own
        sw = mods["pso"].Swarm("/tmp/gdpso_test", disc, n, t, d, torch.device("cuda:0"), tolerance=float(g["tolerance"]),
                               w_inertia=g["w"][0], w_cogn=g["w"][1], w_soci=g["w"][2])
        sw.verbose, sw.record_fitness, sw.comm = False, True, comm
        # separate processes seeded alike draw the same numbers from their global streams; threads share one
        # stream, so every logical rank draws from a stream of its own with that seed
        sw.rng = np.random.RandomState(int(g["np_seed"]))
        return sw, sw.optimize(schedule_inertia=bool(g["schedule_inertia"]), early_stopping=bool(g["early_stopping"]))

    out = run_ranks(world, rank_fn)
    iters = g["fitness"].shape[0]
    safe = first_ambiguous_iteration(g["fitness"], 2e-5)
    assert safe >= 35
    for sw, (history, hp, hv, i_ret) in out:
        res = sw.results
        assert res["hist_pos"].shape[1] == n and res["fitness"].shape[1] == n  # every rank holds the whole swarm's history
        np.testing.assert_array_equal(res["hist_pos"][0], g["pos"][0])
        np.testing.assert_allclose(res["fitness"][:safe], g["fitness"][:safe], atol=TOL)
        np.testing.assert_array_equal(res["hist_pos"][:safe + 1], g["pos"][:safe + 1])
        np.testing.assert_array_equal(res["hist_vel"][:safe + 1], g["vel"][:safe + 1])
        np.testing.assert_allclose(res["mean_mse"][:safe], g["mean_mse"][:safe], rtol=1e-12)
        np.testing.assert_allclose(res["g_best_val_dummy"][:safe], g["g_best_val_dummy"][:safe], atol=TOL)
        assert len(hp) == n and hp["particle_0"].shape[1] == d
        assert len(history["mean_mse"]) == res["iters"] <= iters or res["iters"] == t


def test_swarm_optimize_over_one_rank_rccl_inverter_variant(mods, synth):
    """Swarm.optimize through a real RCCL group (one rank: all this box has) in the inverter variant: encoder-seeded
    positions, targets[lo:hi], MSE-augmented fitness, float32 first move, history gathered over the group."""
    import torch.distributed as dist
    from test_gpu_swarm import build_discovery
    g = load_golden("inverter_out_n8_t5")
    dev = torch.device("cuda:0")
    disc = build_discovery(mods, synth, g, mode="optimize_out_training")
    enc = mods["inv"].Encoder(int(g["z_dim"]), 1).to(dev)
    enc.load_state_dict(synth.to_torch(synth.encoder_state(int(g["z_dim"]), 1, 64, seed=int(g["enc_seed"]))))
    targets = torch.from_numpy(g["targets"])
    n, t, d = int(g["n"]), int(g["t"]), int(g["d"])
    loader = [(targets[k:k + 1], 0) for k in range(n)]
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        np.random.seed(int(g["np_seed"]))
        sw = mods["pso"].Swarm("/tmp/gdpso_test", disc, n, t, d, dev, tolerance=float(g["tolerance"]), w_inertia=g["w"][0],
                               w_cogn=g["w"][1], w_soci=g["w"][2])
        sw.verbose, sw.record_fitness = False, True
        sw.comm = mods["eng"].TorchComm()
        assert not sw.comm.host_staged
        _, hp, _, i_ret = sw.optimize(schedule_inertia=False, early_stopping=False, ood_patient=loader, encoder=enc)
    finally:
        dist.destroy_process_group()
    res = sw.results
    assert i_ret == int(g["i_ret"])
    np.testing.assert_allclose(res["fitness"], g["fitness"], atol=TOL)
    np.testing.assert_allclose(res["hist_pos"], g["pos"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(res["hist_vel"][0], g["vel"][0])
    np.testing.assert_allclose(np.concatenate(sw.g_best_val), g["g_best_val"], atol=TOL)
    np.testing.assert_allclose(res["mean_mse"], g["mean_mse"], rtol=1e-4)


@pytest.mark.parametrize("name", ["inverter_in_n8_t5", "swarm_n12_t40_z10_es"])
def test_swarm_optimize_two_processes_one_gpu(tmp_path, name):
    """Two real processes (gloo rendezvous, collectives staged through the host) sharing this GPU, each running
    Swarm.optimize on its half of the particles: the inverter variant (4 + 4 targets) and an early-stopped run
    (6 + 6 particles, the stop decided identically on both ranks)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GDPSO_CONV_CHAIN="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "sharded_gpu_worker.py"), name, str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-3000:]
    for rank in range(2):
        assert (tmp_path / f"ok{rank}").exists(), res.stderr[-3000:]
