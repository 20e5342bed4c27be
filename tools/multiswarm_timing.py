"""8 swarms x 32 particles (the reference's per-class runs, configs/dcgan_mnist.yaml): one batched run
(util_pso.optimize_swarms) against eight runs one after the other.  Prints one JSON line."""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
imp = lambda m: importlib.import_module("gan-discovery-pso_amd." + m)
synth, pso = imp("synth"), imp("pso.util_pso")
dev = torch.device("cuda:0")
iid = [0, 2, 3, 4, 6, 7, 8, 9]
n, t, d = 32, 50, 10
gen = imp("utils.util_dcgan").Generator(d, 1, 64).to(dev)
gen.load_state_dict(synth.to_torch(synth.generator_state(d, 1, 64, seed=11)))
cnn = imp("pso.util_cnn").get_cnn("ResNet50", 1, iid, 8).to(dev)
cnn.load_state_dict(synth.to_torch(synth.resnet_state("ResNet50", 1, 8, seed=12)))


def make():
    out = []
    for label in iid:
        disc = imp("pso.util_discovery").Discovery(label, gen, cnn, dev, iid_classes=iid)
        sw = pso.Swarm("/tmp/gdpso_t", disc, n, t, d, dev, tolerance=1e-5, w_inertia=0.73, w_cogn=1.496, w_soci=1.496)
        sw.verbose = False
        sw.device_rng_seed = 7
        out.append(sw)
    for sw in out[1:]:
        sw.discovery.share_evaluator_with(out[0].discovery)
    return out


res = {}
for mode in ("sequential", "batched", "sequential", "batched"):
    sws = make()
    np.random.seed(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "batched":
        pso.optimize_swarms(sws, schedule_inertia=False, early_stopping=False)
    else:
        for sw in sws:
            sw.optimize(schedule_inertia=False, early_stopping=False)
    torch.cuda.synchronize()
    res[mode] = time.perf_counter() - t0  # second round of each mode overwrites the warm-up round
print(json.dumps({"swarms": len(iid), "particles_per_swarm": n, "iterations": t,
                  "sequential_s": round(res["sequential"], 4), "batched_s": round(res["batched"], 4),
                  "sequential_evals_per_s": round(len(iid) * n * t / res["sequential"]),
                  "batched_evals_per_s": round(len(iid) * n * t / res["batched"]),
                  "note": "wall time of the optimize calls, history read-back and DataFrame export included"}))
