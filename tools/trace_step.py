#!/usr/bin/env python3
"""Print the per-kernel timeline of a complete swarm iteration (generator, assessor AND the PSO step;
the middle one of those found) in a rocprofv3 --kernel-trace CSV, plus per-kernel totals for it.

    python tools/trace_step.py gpurun_out/prof/<host>/<pid>_kernel_trace.csv [--full]
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    full = "--full" in sys.argv
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # an iteration begins with the generator's first kernel (latents in): gen_resident_kernel (the whole generator), or
    # gen_first_kernel, or latent_prep ahead of a first layer on the matrix cores (gen_resident_kernel then follows
    # within two launches and is not a beginning)
    idx = []
    for i, r in enumerate(rows):
        k = r["Kernel_Name"]
        if "latent_prep" in k or "gen_first_kernel" in k:
            idx.append(i)
        elif "gen_resident" in k and not any("latent_prep" in q["Kernel_Name"] for q in rows[max(0, i - 2):i]):
            idx.append(i)
    if len(idx) < 2:
        raise SystemExit("fewer than two iterations in the trace")
    # the stretches between two generator passes that hold a PSO step (the benchmark ends with passes of
    # generator and assessor alone); of those the middle one, well inside the timed loop
    whole = [(a, b) for a, b in zip(idx[:-1], idx[1:])
             if any("swarm_step" in r["Kernel_Name"] or "move_kernel" in r["Kernel_Name"] for r in rows[a:b])]
    if not whole:
        raise SystemExit("no complete swarm iteration in the trace")
    s, e = whole[len(whole) // 2]
    t0 = int(rows[s]["Start_Timestamp"])
    prev_end = None
    tot = defaultdict(lambda: [0, 0.0])
    busy = 0.0
    for r in rows[s:e]:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gdpso::", "")
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (st - prev_end) / 1e3 if prev_end else 0.0
        prev_end = en
        grid = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Z"]))
        tot[name][0] += 1
        tot[name][1] += (en - st) / 1e3
        busy += (en - st) / 1e3
        if full:
            print(f"{(st - t0) / 1e3:9.1f}us dur {(en - st) / 1e3:7.1f}us gap {gap:5.1f} {name[:36]:36s} grid {grid}")
    span = (int(rows[e]["Start_Timestamp"]) - t0) / 1e3
    print(f"iteration span {span:.1f} us, kernels busy {busy:.1f} us, {e - s} launches")
    for name, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {t:8.1f} us  {c:3d} x  {name}")


if __name__ == "__main__":
    main()
