#!/usr/bin/env python3
"""L2 hits and misses per launch and kernel from one rocprofv3 counter pass (TCC_HIT_sum, TCC_MISS_sum):

    python tools/l2_summary.py <counter_collection.csv> > profiles/rN_l2_hit.txt
"""
import collections
import csv
import sys


def main():
    hit, miss, n = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gdpso::", "").strip()
        if r["Counter_Name"] == "TCC_HIT_sum":
            hit[k] += float(r["Counter_Value"])
            n[k] += 1
        elif r["Counter_Name"] == "TCC_MISS_sum":
            miss[k] += float(r["Counter_Value"])
    print("rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shape")
    print("L2 requests per launch (128-byte lines), summed over the 8 XCDs' L2s:\n")
    print(f"{'kernel':50s} {'launches':>8s} {'hits/launch':>13s} {'misses/launch':>14s} {'hit rate':>9s}")
    for k in sorted(hit, key=lambda k: -(hit[k] + miss[k])):
        if n[k] == 0 or hit[k] + miss[k] == 0:
            continue
        print(f"{k[:50]:50s} {n[k]:8d} {hit[k] / n[k]:13,.0f} {miss[k] / n[k]:14,.0f} {hit[k] / (hit[k] + miss[k]):9.3f}")


if __name__ == "__main__":
    main()
