#!/usr/bin/env python3
"""Experiment: one 512-image evaluation on one stream vs two 256-image halves on two streams
(generator + assessor forward only).  python tools/two_stream_probe.py [n] [iters] [dtype] [z_dim]"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("gan-discovery-pso_amd.engine")
synth = importlib.import_module("gan-discovery-pso_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
zd = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
gen = eng.PackedGenerator(synth.generator_state(zd, 1, 64, seed=11), dev, dtype=dtype)
ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=gen.side, dtype=dtype)
spec = eng.FitnessSpec(class_index=6)
z = torch.randn(n, zd, dtype=torch.float64, device=dev)

def bench(parts):
    evs = [eng.Evaluator(gen, ass) for _ in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    chunks = list(z.chunk(parts))
    def step():
        for ev, st, c in zip(evs, streams, chunks):
            with torch.cuda.stream(st):
                ev.fitness(c, spec)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

for parts in (1, 2, 1, 2, 4):
    print(f"{parts} stream(s) x {n // parts} images: {bench(parts):.3f} ms per evaluation of {n}", flush=True)
