#!/usr/bin/env python3
"""Per-iteration time of the benchmark's fused loop from the first iteration of a process on (HIP events
between iterations enqueued back to back): shows how many iterations the device takes to reach its steady rate.

    python tools/iter_ramp.py [iterations]
"""
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
eng = importlib.import_module("gan-discovery-pso_amd.engine")
synth = importlib.import_module("gan-discovery-pso_amd.synth")


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    dev = torch.device("cuda:0")
    gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11, first_kernel=7), dev)
    ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=gen.side)
    ev = eng.Evaluator(gen, ass)
    spec = eng.FitnessSpec(class_index=6)
    p = 512
    sw = eng.HipSwarm(ev, p, 10, iters + 1, history=True, diagnostics=True)
    rs = np.random.RandomState(42)
    r = rs.standard_normal((p, 2, 10))
    sw.load(r[:, 0], (r[:, 1] - 0.5) / 10, 0.73)
    import time
    sw2 = eng.HipSwarm(ev, p, 10, 4 * iters + 1, history=False, diagnostics=True)
    sw2.load(r[:, 0], (r[:, 1] - 0.5) / 10, 0.73)
    first = 1

    def run(label, before=None):
        nonlocal first
        if before:
            before()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        torch.cuda.synchronize()
        marks[0].record()
        for i in range(first, first + iters):
            sw2.iterate(i, 1, spec, 1.496, 1.496, False, False, 1e-5, True, seed=42)
            marks[i - first + 1].record()
        torch.cuda.synchronize()
        first += iters
        t = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(iters)])
        print(label, flush=True)
        for lo, hi in ((0, 1), (1, 5), (5, 10), (10, 25), (25, 50), (50, 100), (100, 200), (200, 400)):
            if hi <= iters:
                print(f"  iterations {lo + 1:4d}..{hi:4d}: {t[lo:hi].mean():.4f} ms each", flush=True)

    run("from the start of the process")
    run("again, right behind")
    run("after 1 s of idling", lambda: time.sleep(1.0))

    def matmuls():
        a = torch.randn(8192, 8192, device=dev)
        for _ in range(20):
            a = (a @ a) * 1e-4
        torch.cuda.synchronize()
        time.sleep(0.0)

    run("after 1 s of idling and ~0.2 s of fp32 matrix products", lambda: (time.sleep(1.0), matmuls()))


if __name__ == "__main__":
    main()
