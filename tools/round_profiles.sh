#!/bin/bash
# Everything under profiles/ for one round, on a GPU box (from the repository root):
#   bash tools/round_profiles.sh r3
# needs tools/libgdpso_stamps.so (python tools/stamp_probe.py build, on the build machine).
# Results: gpurun_out/round_<tag>/ (copy into profiles/).  The first failing step ends the script.
set -e
TAG=${1:-r3}
OUT=gpurun_out/round_${TAG}
rm -rf $OUT && mkdir -p $OUT
export TMPDIR=/tmp
bash tools/refresh_profiles.sh ${TAG}
cp gpurun_out/refresh_${TAG}/final/* $OUT/
bash tools/refresh_profiles.sh ${TAG}_bf16 --dtype bf16 --z-dim 100
cp gpurun_out/refresh_${TAG}_bf16/final/* $OUT/
timeout -k 10 200 python3 tools/chain_probe.py 512 6 f32 > $OUT/${TAG}_chain_steps.txt 2> $OUT/chain_f32.err
echo "chain probe f32 done"
timeout -k 10 200 python3 tools/chain_probe.py 512 6 bf16 > $OUT/${TAG}_bf16_chain_steps.txt 2> $OUT/chain_bf16.err
echo "chain probe bf16 done"
timeout -k 10 200 python3 tools/iter_ramp.py 100 > $OUT/${TAG}_iter_ramp.txt 2> $OUT/iter_ramp.err
echo "iteration ramp done"
GDPSO_BENCH_FORCE_COMM=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench_one_rank_rccl.json 2> $OUT/bench_comm.err
echo "one-rank RCCL bench done"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --dtype f16 --side 64 > $OUT/${TAG}_f16_64_bench.json 2> $OUT/bench_f16.err
echo "fp16 64x64 bench done"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-shape > $OUT/pmc_l2.log 2>&1
echo "L2 pass done"
python3 tools/l2_summary.py $(find $OUT/pmc_l2 -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_l2_hit.txt
ls -la $OUT
