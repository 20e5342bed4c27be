"""Worker for tests/test_gpu_sharded_networks.py::test_swarm_optimize_two_processes_one_gpu: one of two
processes that share one GPU (gloo rendezvous; the collectives are staged through the host), running
Swarm.optimize on its block of the particles of a recorded reference run and checking what it returns
(every rank returns the whole swarm's result) against the recording."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_utils import first_ambiguous_iteration  # noqa: E402

TOL = 1e-4


def main():
    name, out_dir = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    imp = lambda n: importlib.import_module("gan-discovery-pso_amd." + n)
    synth, pso, discovery, cnn_mod, dcgan, inv = imp("synth"), imp("pso.util_pso"), imp("pso.util_discovery"), imp("pso.util_cnn"), \
        imp("utils.util_dcgan"), imp("inverter.utils_ae.util_inverter")
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    zd, nc = int(g["z_dim"]), int(g["n_class"])
    gen = dcgan.Generator(zd, 1, 64).to(dev)
    gen.load_state_dict(synth.to_torch(synth.generator_state(zd, 1, 64, seed=int(g["gen_seed"]))))
    iid = g["iid"].tolist() if "iid" in g.files else [0, 1]
    cnn = cnn_mod.get_cnn("ResNet50", 1, iid, nc).to(dev)
    cnn.load_state_dict(synth.to_torch(synth.resnet_state("ResNet50", 1, nc, seed=int(g["cnn_seed"]))))
    inverter = name.startswith("inverter")
    mode = "optimize_in_training" if name.startswith("inverter_in") else "optimize_out_training"
    disc = discovery.Discovery(int(g["label"]) if "label" in g.files else 5, gen, cnn, dev, control_pso_fitness=mode,
                               iid_classes=iid if "iid" in g.files else None, obj_fun_threshold=0.0)
    n, t, d = int(g["n"]), int(g["t"]), int(g["d"])
    np.random.seed(int(g["np_seed"]))  # every process seeds alike, as a launcher running the same script would
    sw = pso.Swarm(os.path.join(out_dir, "plots"), disc, n, t, d, dev, tolerance=float(g["tolerance"]), w_inertia=g["w"][0],
                   w_cogn=g["w"][1], w_soci=g["w"][2])
    sw.verbose, sw.record_fitness = False, True
    kw = {}
    if inverter:
        enc = inv.Encoder(zd, 1).to(dev)
        enc.load_state_dict(synth.to_torch(synth.encoder_state(zd, 1, 64, seed=int(g["enc_seed"]))))
        targets = torch.from_numpy(g["targets"])
        kw = dict(ood_patient=[(targets[k:k + 1], 0) for k in range(n)], encoder=enc)
    _, hp, _, i_ret = sw.optimize(schedule_inertia=bool(g["schedule_inertia"]) if "schedule_inertia" in g.files else False,
                                  early_stopping=bool(g["early_stopping"]) if "early_stopping" in g.files else False, **kw)
    res = sw.results
    assert sw._comm() is not None and sw._comm().world == 2 and sw._comm().host_staged
    iters = g["fitness"].shape[0]
    assert res["hist_pos"].shape[1] == n and len(hp) == n  # the whole swarm on every rank
    if inverter:
        assert i_ret == int(g["i_ret"])
        np.testing.assert_allclose(res["fitness"], g["fitness"], atol=TOL)
        np.testing.assert_allclose(res["hist_pos"], g["pos"], rtol=1e-4, atol=1e-4)
        np.testing.assert_array_equal(res["hist_vel"][0], g["vel"][0])
        np.testing.assert_allclose(np.concatenate(sw.g_best_val), g["g_best_val"], atol=TOL)
    else:
        safe = first_ambiguous_iteration(g["fitness"], 2e-5)
        assert safe >= 5
        np.testing.assert_allclose(res["fitness"][:safe], g["fitness"][:safe], atol=TOL)
        np.testing.assert_array_equal(res["hist_pos"][:safe + 1], g["pos"][:safe + 1])
        np.testing.assert_allclose(res["mean_mse"][:safe], g["mean_mse"][:safe], rtol=1e-12)
        if safe == iters:
            assert i_ret == int(g["i_ret"])
            np.testing.assert_array_equal(sw.g_best_pos, g["g_best_pos"])
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok\n")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
