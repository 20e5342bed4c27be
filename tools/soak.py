#!/usr/bin/env python3
"""Soak run of the fused loop: the benchmark's swarm (512 particles, ResNet50, 28x28) for many iterations in chunks,
twice from the same start; reports the rate, that every recorded value is finite, that the chained launch never
raised its abort flag, and that the two runs left the same bits (the loop is deterministic).

    python tools/soak.py [iterations] [particles] [dtype] > gpurun_out/soak.json
"""
import importlib
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
eng = importlib.import_module("gan-discovery-pso_amd.engine")
synth = importlib.import_module("gan-discovery-pso_amd.synth")


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
    chunk = 64
    dev = torch.device("cuda:0")
    gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11), dev, dtype=dtype)
    ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=gen.side, dtype=dtype)
    ev = eng.Evaluator(gen, ass)
    spec = eng.FitnessSpec(class_index=6)
    r = np.random.RandomState(42).standard_normal((p, 2, 10))
    runs = []
    for rep in range(2):
        sw = eng.HipSwarm(ev, p, 10, iters, history=False, diagnostics=True)
        sw.load(r[:, 0], (r[:, 1] - 0.5) / 10, 0.73)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = 1
        while first <= iters:
            count = min(chunk, iters - first + 1)
            sw.iterate(first, count, spec, 1.496, 1.496, True, False, 1e-5, True, seed=42)
            first += count
            if (first // chunk) % 32 == 0:
                eng.check_chains(dev)  # (synchronises: the abort flag of the chained launch)
                print(f"run {rep}: {first - 1} iterations", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.check_chains(dev)
        res = sw.results()
        runs.append((dt, res))
    a, b = runs[0][1], runs[1][1]
    same = all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in ("pos", "vel", "pbest_val", "g_best_pos", "mean_mse", "g_best_val"))
    finite = all(np.isfinite(np.asarray(a[k], dtype=np.float64)).all() for k in ("pos", "vel", "g_best_pos", "mean_mse", "g_best_val"))
    print(json.dumps({"iterations": iters, "particles": p, "dtype": dtype, "chunk": chunk,
                      "particle_evals_per_s": [round(p * iters / dt, 1) for dt, _ in runs],
                      "ms_per_iteration": [round(1e3 * dt / iters, 4) for dt, _ in runs],
                      "iterations_done": int(a["iters"]), "all_finite": bool(finite), "two_runs_same_bits": bool(same),
                      "final_gbest_val": float(a["gbest_val"]), "chain_aborts": 0}))


if __name__ == "__main__":
    main()
