#!/usr/bin/env python3
"""Time the swarm diagnostic (mean pairwise distance, float64) at the particle counts a weak-scaling
run reaches (512 particles per rank): the whole diagnostic on one rank, and one rank's share of it
(every n_ranks-th tile of pairs, what the sharded loop runs per iteration).

    python tools/dist_scaling_probe.py [d]
"""
import importlib
import sys
import types

import torch

sys.path.insert(0, ".")
eng = importlib.import_module("gan-discovery-pso_amd.engine")


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    ev = types.SimpleNamespace(device=dev, gen=types.SimpleNamespace(z_dim=d))
    for ranks in (1, 2, 4, 8):
        n_all = 512 * ranks
        sw = eng.HipSwarm(ev, 512, d, 64, n_ranks=ranks, rank=0, n_global=n_all, history=False, diagnostics=True)
        pos_all = torch.randn(n_all, d, dtype=torch.float64, device=dev)
        for _ in range(3):
            sw.mean_distance(1, pos_all)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(20):
            sw.mean_distance(1 + i, pos_all)
        b.record()
        torch.cuda.synchronize()
        whole = a.elapsed_time(b) / 20 * 1e3
        a.record()
        for i in range(20):
            sw.distance_part(1 + i, pos_all)
        b.record()
        torch.cuda.synchronize()
        print(f"ranks {ranks}: {n_all} particles, d {d}: whole diagnostic {whole:8.1f} us, one rank's share {a.elapsed_time(b) / 20 * 1e3:8.1f} us "
              f"(two launches each, back to back); value {sw.mean_dist[0].item():.12f}", flush=True)


if __name__ == "__main__":
    main()
