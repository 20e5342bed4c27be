#!/bin/bash
# Regenerates the measurement files under profiles/ on a GPU box (run from the repository root):
#   bash tools/refresh_profiles.sh r1
# Steps are chained: the first failing step ends the script.  Counter passes run on their own,
# one counter each, with the program directly after "--".
set -e
TAG=${1:-r1}
shift || true
ARGS="$@"                     # extra bench.py arguments, e.g. --dtype bf16 --z-dim 100
DTYPE=f32
case "$ARGS" in *bf16*) DTYPE=bf16;; *f16*) DTYPE=f16;; esac
OUT=gpurun_out/refresh_${TAG}
rm -rf $OUT && mkdir -p $OUT
export TMPDIR=/tmp

timeout -k 10 300 python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-cpu-baseline --no-reference-shape $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
echo "kernel trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shape $ARGS > $OUT/pmc_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shape $ARGS > $OUT/pmc_write.log 2>&1
echo "write pass done"
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shape $ARGS > $OUT/pmc_mfma.log 2>&1
echo "mfma pass done"

STATS=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
TRACE=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
FETCH=$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1)
WRITE=$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
MFMA=$(find $OUT/pmc_mfma -name '*counter_collection.csv' | head -1)
mkdir -p $OUT/final
cp $OUT/bench.json $OUT/final/${TAG}_bench.json
cp $OUT/bench_under_rocprof.json $OUT/final/${TAG}_bench_under_rocprof.json
cp "$STATS" $OUT/final/${TAG}_bench_kernel_stats.csv
python3 tools/trace_step.py "$TRACE" --full > $OUT/final/${TAG}_iteration_timeline.txt
python3 tools/pmc_summary.py "$FETCH" "$WRITE" > $OUT/final/${TAG}_hbm_traffic.json
python3 tools/mfma_util.py "$MFMA" "$STATS" "$TRACE" --dtype $DTYPE > $OUT/final/${TAG}_mfma_util.json
ls -la $OUT/final
