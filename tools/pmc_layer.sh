#!/bin/bash
# Hardware counters of ONE convolution layer (gdpso_conv_probe), one pass:
#   bash tools/pmc_layer.sh <tag> "<n hin cin cout k stride pad tile splits res>" COUNTER [COUNTER ...]
# (at most 8 SQ_, 4 TCP_, 2 TA_ counters in a pass).  Prints per-launch averages of the conv kernel.
set -e
TAG=$1; SHAPE=$2; shift 2
OUT=gpurun_out/pmc_layer_$TAG
rm -rf $OUT
export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 tools/probe_one.py $SHAPE 20 > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if "conv_gemm" not in r["Kernel_Name"]:
        continue
    a = acc[r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:32s} {v / n:16.0f} per launch ({n} launches)")
PY
