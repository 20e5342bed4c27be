# The resident launches against the tiled kernels at batches of several rounds of workgroups.
for n in ${SIZES:-768 1536 2048 3072 4096}; do
  for v in "GDPSO_STAGE1_RESIDENT=0" "GDPSO_STAGE1_RESIDENT_MAX=65536"; do
    BENCH_ARGS="--particles $n --steps 20" bash tools/ab_bench.sh "$v" | sed "s/^/n=$n /"
  done
done
