#!/usr/bin/env python3
"""First generator layer: the direct kernel (short latents) against the matrix-core path, image by image.
    python tools/gen_first_check.py"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
eng = importlib.import_module("gan-discovery-pso_amd.engine")
synth = importlib.import_module("gan-discovery-pso_amd.synth")
dev = torch.device("cuda:0")
for dtype in ("f32", "bf16", "f16"):
    for n in (512, 37):
        z = torch.from_numpy(np.random.RandomState(3).standard_normal((n, 10))).to(dev)
        out = []
        for direct in ("0", "1"):
            os.environ["GDPSO_GEN_FIRST_DIRECT"] = direct
            gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11), dev, dtype=dtype)
            out.append(gen.forward(z).cpu().numpy())
        d = np.abs(out[0] - out[1])
        print(dtype, n, "max |diff|", d.max(), "images that differ", int((d.reshape(n, -1).max(axis=1) > 0).sum()))
