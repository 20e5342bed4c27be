#!/usr/bin/env python3
"""Chained launches on side streams: one evaluator on a side stream; two evaluators alternating with a
device synchronisation between calls; two evaluators interleaved without one."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("gan-discovery-pso_amd.engine")
synth = importlib.import_module("gan-discovery-pso_amd.synth")
dev = torch.device("cuda:0")
gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11), dev)
ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=28)
evs = [eng.Evaluator(gen, ass), eng.Evaluator(gen, ass)]
z = [torch.from_numpy(np.random.RandomState(s).standard_normal((512, 10))).cuda() for s in (1, 2)]
spec = eng.FitnessSpec(class_index=2)
want = [evs[i].fitness(z[i], spec)[0].cpu().numpy().copy() for i in range(2)]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
torch.cuda.synchronize()


def report(name, got):
    for i, g in enumerate(got):
        if g is None:
            continue
        bad = np.nonzero(g.cpu().numpy() != want[i])[0]
        print(f"{name}: evaluator {i}: {bad.size} of 512 differ; images {bad[:4].tolist()} ... shares hit {sorted(set((bad // 64).tolist()))}")


with torch.cuda.stream(streams[0]):
    g0 = evs[0].fitness(z[0], spec)[0]
torch.cuda.synchronize()
report("one evaluator, side stream", [g0, None])
got = [None, None]
for rep in range(3):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            got[i] = evs[i].fitness(z[i], spec)[0]
        torch.cuda.synchronize()
report("alternating, synchronised", got)
for rep in range(3):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            got[i] = evs[i].fitness(z[i], spec)[0]
torch.cuda.synchronize()
report("interleaved", got)
import time
_lib = importlib.import_module("gan-discovery-pso_amd._lib")
try:
    eng.check_chains()
except Exception as e:
    print(e, "|", _lib.lib().gdpso_last_error())
for rep in range(2):
    for i in range(2):
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[i]):
            got[i] = evs[i].fitness(z[i], spec)[0]
        print(f"rep {rep} evaluator {i}: call returned after {1e3 * (time.perf_counter() - t0):.1f} ms")
t0 = time.perf_counter()
torch.cuda.synchronize()
print(f"drained after {1e3 * (time.perf_counter() - t0):.1f} ms")
report("interleaved again", got)
