#!/usr/bin/env python3
"""MFMA pipe utilisation per kernel: MFMA busy cycles from one rocprofv3 PMC pass of bench.py
(--pmc SQ_VALU_MFMA_BUSY_CYCLES) against the kernel durations of the --kernel-trace --stats run.

    python tools/mfma_util.py <counter_collection.csv> <kernel_stats.csv> [<kernel_trace.csv>] [--dtype f32|bf16|f16]
        > profiles/rN_mfma_util.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs.  fp32: 64 per
v_mfma_f32_32x32x2_f32 (4096 FLOP), 32 per v_mfma_f32_16x16x4_f32 (2048 FLOP): 64 FLOP a cycle, so
busy cycles x 64 = FLOPs the matrix pipes really executed.  16-bit: 32 per v_mfma_f32_32x32x16_bf16 /
_f16 (32768 FLOP): 1024 FLOP a cycle.  (The counter pass's own GRBM_GUI_ACTIVE is no use as the
denominator: it includes ~9 us of per-dispatch serialisation that a normal run does not have.)
(stage_resident_kernel and gen_resident_kernel with 16-bit storage mix the two: the stem and the generator's last layer
run fp32 MFMAs, the rest 16-bit ones; their rows are priced at the 16-bit rate like the other convolution launches and
so understate the fp32 layers' share.)
utilisation = executed FLOP/s over the peak of the operand type (157.3 / 2500 TFLOP/s at 2.4 GHz; the
chip holds ~2.1 GHz under the fp32 load, so 0.875 is the ceiling there).

With the kernel trace as third argument the two MFMA convolutions of the generator get rows of
their own: the two convolution launches behind every latent_prep_kernel launch (G1: ConvTranspose
z -> 2F on the 1x1 latent, a GEMM; G3: ConvTranspose 2F -> F, k4 s2 p1, four sub-pixel phases).
"""
import collections
import csv
import json
import sys

PEAK = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}
FLOP_PER_BUSY_CYCLE = {"f32": 64.0, "bf16": 1024.0, "f16": 1024.0}


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def generator_rows(rows, key_time, value_of):
    """value_of(row) for the generator's convolution launches: the first and second conv_gemm launch behind a
    latent_prep launch (G1, G3), or the one launch behind gen_first_kernel (G3; the first layer is that kernel)."""
    rows = sorted(rows, key=key_time)
    out = {"G1": [], "G3": []}
    pending = []
    for r in rows:
        k = short(r["Kernel_Name"])
        if "latent_prep" in k:
            pending = ["G1", "G3"]
        elif "gen_first_kernel" in k:
            pending = ["G3"]
        elif pending and "conv_gemm" in k:
            out[pending.pop(0)].append(value_of(r))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dtype = sys.argv[sys.argv.index("--dtype") + 1] if "--dtype" in sys.argv else "f32"
    peak, fpc = PEAK[dtype], FLOP_PER_BUSY_CYCLE[dtype]
    busy = collections.defaultdict(float)
    n = collections.defaultdict(int)
    crow = [r for r in csv.DictReader(open(args[0])) if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"]
    for r in crow:
        busy[short(r["Kernel_Name"])] += float(r["Counter_Value"])
        n[short(r["Kernel_Name"])] += 1
    avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(args[1]))}
    out = {"counter_pass": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline",
           "durations": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline (AverageNs)",
           "dtype": dtype, "peak_tflops": peak, "kernels": {}}
    tot_flop = tot_ns = 0.0
    for k in sorted(busy, key=lambda k: -busy[k]):
        if busy[k] == 0 or k not in avg_ns:
            continue
        # the stem and the last generator layer stay fp32 MFMA whatever the storage type
        f = 64.0 if ("stem" in k or "gen_last" in k) else fpc
        p = 157.3 if ("stem" in k or "gen_last" in k) else peak
        flop = busy[k] / n[k] * f
        tf = flop / avg_ns[k] / 1e3
        out["kernels"][k] = {"launches_counted": n[k], "avg_us": round(avg_ns[k] / 1e3, 2), "executed_gflop_per_launch": round(flop / 1e9, 3),
                             "executed_tflops": round(tf, 1), "mfma_util": round(tf / p, 3)}
        if "conv_gemm" in k or "conv_chain" in k or "stage_resident" in k or "gen_resident" in k:
            tot_flop += flop * n[k]
            tot_ns += avg_ns[k] * n[k]
    out["kernels"]["conv_gemm (all instantiations)"] = {"executed_tflops": round(tot_flop / tot_ns / 1e3, 1),
                                                         "mfma_util": round(tot_flop / tot_ns / 1e3 / peak, 3)}
    if len(args) > 2:
        cyc = generator_rows(crow, lambda r: int(r["Dispatch_Id"]), lambda r: float(r["Counter_Value"]))
        trows = list(csv.DictReader(open(args[2])))
        dur = generator_rows(trows, lambda r: int(r["Start_Timestamp"]), lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        names = {"G1": "generator G1: ConvTranspose2d(Z -> 2F, k, s1, p0) as a GEMM", "G3": "generator G3: ConvTranspose2d(2F -> F, k4, s2, p1)"}
        out["generator_layers"] = {}
        for g in ("G1", "G3"):
            if cyc[g] and dur[g]:
                flop = sum(cyc[g]) / len(cyc[g]) * fpc
                ns = sum(dur[g][2:]) / max(1, len(dur[g][2:]))  # without the first (cold) launches
                out["generator_layers"][names[g]] = {"launches_counted": len(cyc[g]), "avg_us": round(ns / 1e3, 2),
                                                     "executed_gflop_per_launch": round(flop / 1e9, 3),
                                                     "executed_tflops": round(flop / ns / 1e3, 1),
                                                     "mfma_util": round(flop / ns / 1e3 / peak, 3)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
