#!/usr/bin/env python3
"""MFMA pipe utilisation per kernel: MFMA busy cycles from one rocprofv3 PMC pass of bench.py
(--pmc SQ_VALU_MFMA_BUSY_CYCLES) against the kernel durations of the --kernel-trace --stats run.

    python tools/mfma_util.py <counter_collection.csv> <kernel_stats.csv> > profiles/rN_mfma_util.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs, 64 per
v_mfma_f32_32x32x2_f32 (4096 FLOP) and 32 per v_mfma_f32_16x16x4_f32 (2048 FLOP): 64 FLOP a cycle
either way, so busy cycles x 64 = FLOPs the matrix pipes really executed.  (The counter pass's
own GRBM_GUI_ACTIVE is no use as the denominator: it includes ~9 us of per-dispatch
serialisation that a normal run does not have.)  utilisation = executed FLOP/s over the
157.3 TFLOP/s fp32 peak (2.4 GHz; the chip holds ~2.1 GHz under this load, so 0.875 is the ceiling).
"""
import collections
import csv
import json
import sys

PEAK_TFLOPS = 157.3


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    busy = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(sys.argv[1])):
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[short(r["Kernel_Name"])] += float(r["Counter_Value"])
            n[short(r["Kernel_Name"])] += 1
    avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(sys.argv[2]))}
    out = {"counter_pass": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline",
           "durations": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline (AverageNs)",
           "peak_tflops": PEAK_TFLOPS, "kernels": {}}
    tot_flop = tot_ns = 0.0
    for k in sorted(busy, key=lambda k: -busy[k]):
        if busy[k] == 0 or k not in avg_ns:
            continue
        flop = busy[k] / n[k] * 64.0
        tf = flop / avg_ns[k] / 1e3
        out["kernels"][k] = {"launches_counted": n[k], "avg_us": round(avg_ns[k] / 1e3, 2), "executed_gflop_per_launch": round(flop / 1e9, 3),
                             "executed_tflops": round(tf, 1), "mfma_util": round(tf / PEAK_TFLOPS, 3)}
        if "conv_gemm" in k:
            tot_flop += flop * n[k]
            tot_ns += avg_ns[k] * n[k]
    out["kernels"]["conv_gemm (all instantiations)"] = {"executed_tflops": round(tot_flop / tot_ns / 1e3, 1),
                                                         "mfma_util": round(tot_flop / tot_ns / 1e3 / PEAK_TFLOPS, 3)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
