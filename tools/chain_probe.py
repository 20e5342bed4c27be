#!/usr/bin/env python3
"""Where the chained launch spends its time, step by step (diagnostic build, tools/libgdpso_stamps.so:
`python tools/stamp_probe.py build` here, then `python tools/chain_probe.py [n [step [dtype]]]` on the GPU; step: the
one whose tiles also leave the tile code's own stamps, dtype f32 | bf16 | f16).

Per step (over the workgroups of XCD share 0 and over all): when the first / last workgroup entered, how
long the waits lasted, how long the tiles took, and the hand-off: last signal of step s-1 -> last wait
of step s ended."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    eng = importlib.import_module("gan-discovery-pso_amd.engine")
    synth = importlib.import_module("gan-discovery-pso_amd.synth")
    _lib = importlib.import_module("gan-discovery-pso_amd._lib")
    _lib.LIB_PATH = os.path.join(HERE, "libgdpso_stamps.so")  # the diagnostic build instead of the product library
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
    ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, 8, seed=12), dev, img_side=28, dtype=dtype)
    gen = eng.PackedGenerator(synth.generator_state(10, 1, 64, seed=11), dev, dtype=dtype)
    ev = eng.Evaluator(gen, ass)
    x = torch.rand(n, 28, 28, device=dev)
    for _ in range(3):
        ev.assess(x, eng.FitnessSpec(class_index=0))
    buf = torch.zeros(512, 192, 4, dtype=torch.int64, device=dev)
    tile_step = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    tbuf = torch.zeros(8192, 16, dtype=torch.int64, device=dev)  # (the other convolution kernels stamp their workgroups' rows too)
    lib.gdpso_chain_stamps_set.argtypes = [C.c_void_p, C.c_int]
    lib.gdpso_conv_stamps_set.argtypes = [C.c_void_p]
    assert lib.gdpso_conv_stamps_set(tbuf.data_ptr()) == 0
    assert lib.gdpso_chain_stamps_set(buf.data_ptr(), tile_step) == 0
    ev.assess(x, eng.FitnessSpec(class_index=0))
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().astype(np.int64)
    ident = raw[:, 191, 0]
    share = (ident & 0xff) - 1
    s = raw * 0.01  # us
    s[:, 191] = 0
    t0 = s[s > 0].min()
    nsteps = int(np.nonzero((s[:, :, 2] > 0).any(axis=0))[0].max()) + 1
    print(f"{nsteps} steps; us; per step the median over the 8 shares of: workgroups with a tile, wait (enter -> wait ended), "
          "tiles (wait end -> done; median and max over workgroups), drain + signal, hand-off (last signal of the previous step -> "
          "last wait end), step (last signal -> last signal)")
    print(" step  wgs    wait   tiles(med max)  signal  handoff    step")
    tot = 0.0
    for k in range(nsteps):
        rows = []
        for sh in range(16):
            live = (s[:, k, 2] > 0) & (share == sh)
            if not live.any():
                continue
            e, w, d, g = (s[live, k, i] for i in range(4))
            start = np.where(w > 0, w, e)
            prev = s[(share == sh), k - 1, 3].max() if k > 0 else e.min()
            hand = (w.max() - prev) if k > 0 else 0.0
            rows.append((live.sum(), np.median(start - e), np.median(d - start), (d - start).max(), np.median(g - d), hand, g.max() - prev))
        r = np.median(np.array(rows, dtype=float), axis=0)
        tot += r[6]
        print(f" {k:3d}  {int(r[0]):4d}  {r[1]:6.2f}  {r[2]:6.2f} {r[3]:6.2f}  {r[4]:6.2f}  {r[5]:7.2f}  {r[6]:6.2f}")
    print(f"sum of step times {tot:.2f} us; whole chain: {s.max() - t0:.2f} us")
    if tile_step >= 0:
        # who is slow: the step's tile times of share 0 by member (members m and 63 - m share a CU)
        member = (ident >> 8) & 0xffff
        live = (s[:, tile_step, 2] > 0) & (share == 0)
        order = np.argsort(member[live])
        tt = (s[live, tile_step, 2] - np.where(s[live, tile_step, 1] > 0, s[live, tile_step, 1], s[live, tile_step, 0]))[order]
        print(f"step {tile_step}, share 0: tile time by member:", " ".join(f"{v:.1f}" for v in tt))
        done = (s[live, tile_step, 2] - s[live, tile_step, 2].min())[order]
        print(f"step {tile_step}, share 0: finished after the first by:", " ".join(f"{v:.1f}" for v in done))
        # the tile code's own stamps (s_memtime ticks; slots 7 / 6 hold s_memrealtime at entry / exit) of the first tile
        # each workgroup ran in that step
        ts = tbuf.cpu().numpy().astype(np.int64)
        # per workgroup: its own signal of the previous step -> entry of this step -> entry of the tile code
        # (the step's arguments fetched in between), on the 100 MHz clock
        both = (ts[:512, 11] != 0) & (s[:, tile_step, 0] > 0) & (s[:, tile_step - 1, 3] > 0)
        gap = s[both, tile_step, 0] - s[both, tile_step - 1, 3]
        args = ts[:512, 7][both] * 0.01 - s[both, tile_step, 0]
        whole = np.where(s[both, tile_step, 1] > 0, s[both, tile_step, 1], s[both, tile_step, 0]) - s[both, tile_step - 1, 3]
        print(f"step {tile_step}: own signal of step {tile_step - 1} -> step entry {np.median(gap):.2f} ({np.percentile(gap, 90):.2f}); "
              f"step entry -> tile code entered {np.median(args):.2f} ({np.percentile(args, 90):.2f}); "
              f"own signal -> own wait ended {np.median(whole):.2f} ({np.percentile(whole, 90):.2f})")
        ts = ts[ts[:, 11] != 0]  # rows stamped by the chain (only it has slot 11)
        ghz = float(np.median((ts[:, 3] - ts[:, 0]) / np.maximum(1, (ts[:, 6] - ts[:, 7]) * 10.0)))
        print(f"step {tile_step}: {ts.shape[0]} workgroups stamped, in-kernel clock ~{ghz:.2f} GHz; medians (p90) in us:")
        for name, a, b in (("entry -> phase header in SGPRs", 0, 8), ("-> weights requested (wait begins)", 8, 11), ("wait", 11, 12),
                           ("-> activations requested", 12, 9), ("-> returned", 9, 10), ("-> LDS stores + barrier", 10, 1),
                           ("K loop", 1, 2), ("epilogue -> stores drained", 2, 3)):
            v = (ts[:, b] - ts[:, a]) / (ghz * 1e3)
            print(f"   {name:38s} {np.median(v):6.2f} ({np.percentile(v, 90):6.2f})")


if __name__ == "__main__":
    main()
