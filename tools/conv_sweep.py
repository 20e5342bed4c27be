#!/usr/bin/env python3
"""Sweep (tile, split-K) for every distinct convolution shape of the ResNet50 assessor and
the generator at a given batch, using gdpso_conv_probe.  Prints TFLOP/s (nominal FLOPs).

    python tools/conv_sweep.py [--n 512] [--quick]
"""
import argparse
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("gan-discovery-pso_amd._lib")

# (name, hin, cin, cout, k, stride, pad, with_res, count per forward)
RESNET50_28 = [
    ("s1.c1a", 7, 64, 64, 1, 1, 0, 0, 1), ("s1.c1b", 7, 256, 64, 1, 1, 0, 0, 2), ("s1.c2", 7, 64, 64, 3, 1, 1, 0, 3),
    ("s1.c3", 7, 64, 256, 1, 1, 0, 1, 3), ("s1.ds", 7, 64, 256, 1, 1, 0, 0, 1),
    ("s2.c1a", 7, 256, 128, 1, 1, 0, 0, 1), ("s2.c2a", 7, 128, 128, 3, 2, 1, 0, 1), ("s2.ds", 7, 256, 512, 1, 2, 0, 0, 1),
    ("s2.c1b", 4, 512, 128, 1, 1, 0, 0, 3), ("s2.c2b", 4, 128, 128, 3, 1, 1, 0, 3), ("s2.c3", 4, 128, 512, 1, 1, 0, 1, 4),
    ("s3.c1a", 4, 512, 256, 1, 1, 0, 0, 1), ("s3.c2a", 4, 256, 256, 3, 2, 1, 0, 1), ("s3.ds", 4, 512, 1024, 1, 2, 0, 0, 1),
    ("s3.c1b", 2, 1024, 256, 1, 1, 0, 0, 5), ("s3.c2b", 2, 256, 256, 3, 1, 1, 0, 5), ("s3.c3", 2, 256, 1024, 1, 1, 0, 1, 6),
    ("s4.c1a", 2, 1024, 512, 1, 1, 0, 0, 1), ("s4.c2a", 2, 512, 512, 3, 2, 1, 0, 1), ("s4.ds", 2, 1024, 2048, 1, 2, 0, 0, 1),
    ("s4.c1b", 1, 2048, 512, 1, 1, 0, 0, 2), ("s4.c2b", 1, 512, 512, 3, 1, 1, 0, 2), ("s4.c3", 1, 512, 2048, 1, 1, 0, 1, 3),
]


DTYPE = 0


def probe(n, hin, cin, cout, k, stride, pad, tile, splits, res, reps=20):
    us, tu, su = C.c_double(), C.c_int(), C.c_int()
    lib.check(lib.lib().gdpso_conv_probe_typed(n, hin, cin, cout, k, stride, pad, tile, splits, res, reps, C.byref(us), C.byref(tu),
                                               C.byref(su), DTYPE))
    return us.value, tu.value, su.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--dtype", choices=sorted(lib.DTYPES), default="f32")
    args = ap.parse_args()
    global DTYPE
    DTYPE = lib.DTYPES[args.dtype]
    kq = 32 if args.dtype == "f32" else 64
    total_plan = total_best = 0.0
    for name, hin, cin, cout, k, stride, pad, res, count in RESNET50_28:
        hout = (hin + 2 * pad - k) // stride + 1
        flops = 2.0 * args.n * hout * hout * cout * cin * k * k
        us0, t0, s0 = probe(args.n, hin, cin, cout, k, stride, pad, 0, 0, res)
        line = f"{name:7s} M={args.n * hout * hout:6d} K={cin * k * k:5d} N={cout:5d}  plan t{t0} s{s0}: {us0:6.1f}us {flops / us0 / 1e6:6.1f}TF |"
        best = (us0, t0, s0)
        if not args.quick:
            for tile in (16, 32, 64, 128):
                if (tile == 16 and cin % (4 * kq)) or (tile == 32 and cin % (2 * kq)) or (tile == 128 and cout % 128):
                    continue
                for splits in (1, 2) if tile == 128 else (1, 2, 4):
                    if splits > 1 and cin * k * k // splits < 64:
                        continue
                    us, _, _ = probe(args.n, hin, cin, cout, k, stride, pad, tile, splits, res)
                    line += f" t{tile}s{splits}:{us:5.1f}"
                    if us < best[0]:
                        best = (us, tile, splits)
        print(line + f" | best t{best[1]} s{best[2]} {best[0]:.1f}us {flops / best[0] / 1e6:.1f}TF", flush=True)
        total_plan += us0 * count
        total_best += best[0] * count
    print(f"sum over the forward pass: planner {total_plan:.0f} us, best {total_best:.0f} us")


if __name__ == "__main__":
    main()
