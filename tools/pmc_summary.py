#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rN_hbm_traffic.json

Counter values are KiB.  FETCH_SIZE is doubled: gfx950 tallies the 128-byte requests of
16-byte-per-lane reads at 64 bytes.  WRITE_SIZE is taken as read.
"""
import collections
import csv
import json
import sys


def agg(name, path):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        a[k][0] += 1
        a[k][1] += float(r["Counter_Value"])
    return a


def main():
    fe, wr = agg("FETCH_SIZE", sys.argv[1]), agg("WRITE_SIZE", sys.argv[2])
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one pass each) -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline",
           "corrections": "counter values are KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests of 16-B-per-lane reads at 64 B); WRITE_SIZE as read",
           "kernels": {}}
    # every convolution launch (tile shapes, paired launches, the chained launch, the resident stages and generator) as one entry
    for a in (fe, wr):
        tot = [0, 0.0]
        for k in list(a):
            if "conv_gemm" in k or "conv_chain" in k or "stage_resident" in k or "gen_resident" in k:
                tot[0] += a[k][0]
                tot[1] += a[k][1]
        a["conv_gemm (all instantiations)"] = tot
    for k in fe:
        if fe[k][1] <= 0 or not (k.startswith("gdpso::") or k.startswith("conv_gemm") or k.startswith("conv_chain") or "stage_resident" in k or "gen_resident" in k):
            continue
        c, v = fe[k]
        cw, vw = wr.get(k, [c, 0.0])
        f, w = 2 * 1024 * v / c, 1024 * vw / max(cw, 1)
        out["kernels"][k] = {"launches": c, "fetch_bytes_per_launch": round(f), "write_bytes_per_launch": round(w),
                             "hbm_bytes_per_launch": round(f + w)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
