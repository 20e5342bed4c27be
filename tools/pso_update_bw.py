#!/usr/bin/env python3
"""Achieved HBM bandwidth of the PSO update kernels at swarm sizes large enough to leave the caches:
move_kernel (update_velocity + update_position, reference src/pso/util_pso.py:43-50, 39-41) and
pbest_kernel + candidate_kernel (update_personal_best's compare-and-copy and this rank's min-loc,
:122-124), N = 2^14 .. 2^20 particles, D = 10 and 100, float64 state like the reference's.

    python tools/pso_update_bw.py > profiles/rN_pso_update_bw.json

Algorithmic bytes per particle (each array once):
  move : read pos, vel, pbest_pos (3 D doubles) + (r1, r2) 16 B; write pos, vel (2 D doubles)  = 40 D + 16
         (+ 16 D for the two history rows when history is kept)
  pbest: read fitness, pbest_val 8 B; write pbest_val 4 B where improved; pbest_pos <- pos for the
         improved particles (16 D each, a fraction p of them)                                 = 8 + p (4 + 16 D)
Times are HIP-event averages over back-to-back launches on the stream the kernels run on; the
fitness values are redrawn between repetitions of pbest so that about half of the particles improve.
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("gan-discovery-pso_amd.engine")
PEAK = 8000.0  # GB/s, HBM3E specification (about 6300 achievable by a plain copy)


def swarm(n, d, history):
    ev = type("E", (), {"device": torch.device("cuda:0"), "gen": type("G", (), {"z_dim": d, "side": 28})()})()
    s = object.__new__(eng.HipSwarm)
    eng.HipSwarm.__init__(s, ev, n, d, 4, history=history, diagnostics=False)
    g = torch.Generator(device="cuda").manual_seed(1)
    s.pos.copy_(torch.randn(n, d, generator=g, device="cuda", dtype=torch.float64))
    s.vel.copy_(torch.randn(n, d, generator=g, device="cuda", dtype=torch.float64) * 0.1)
    s.load(s.pos.cpu().numpy(), s.vel.cpu().numpy(), 0.73)
    return s


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    rows = []
    for d in (10, 100):
        for lg in range(14, 21, 2 if d == 100 else 1):
            n = 1 << lg
            for history in (False, True):
                if history and n * d * 8 * 5 * 2 > 40e9:
                    continue
                s = swarm(n, d, history)
                t_move = timed(lambda: s.move(1, 1.496, 1.496, seed=3), 20)
                b_move = n * (40 * d + 16 + (16 * d if history else 0))
                # pbest: alternate two fitness vectors so that every launch finds improvements
                s.pbest_val.fill_(0.5)
                fit = torch.rand(n, device="cuda", dtype=torch.float32)
                s.fitness.copy_(fit)
                improved = float((fit < 0.5).float().mean())

                def pb():
                    s.pbest_val.fill_(0.5)
                    s.local_best(1)

                t_fill = timed(lambda: s.pbest_val.fill_(0.5), 20)
                t_pb = timed(pb, 20) - t_fill
                b_pb = n * (8 + improved * (4 + 16 * d))
                rows.append({"particles": n, "dim_space": d, "history": history,
                             "move_us": round(t_move * 1e6, 2), "move_GBps": round(b_move / t_move / 1e9, 1),
                             "move_frac_of_8TBps": round(b_move / t_move / 1e9 / PEAK, 4),
                             "pbest_us": round(t_pb * 1e6, 2), "pbest_GBps": round(b_pb / t_pb / 1e9, 1),
                             "pbest_frac_of_8TBps": round(b_pb / t_pb / 1e9 / PEAK, 4), "improved_fraction": round(improved, 3)})
                print(rows[-1], file=sys.stderr, flush=True)
                del s
                torch.cuda.empty_cache()
    json.dump({"what": "PSO update kernels at large swarm sizes: algorithmic bytes / HIP-event time",
               "peak_GBps": PEAK, "state": "float64 positions, velocities, personal bests (the reference's numpy precision); float32 fitness",
               "rows": rows}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
