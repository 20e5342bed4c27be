# The resident launches against the tiled kernels at batches of several rounds of workgroups (on a GPU box, from the
# repository root).  SIZES: particle counts; OLD / NEW: the two environments compared; BENCH_ARGS: more bench.py arguments.
#   bash tools/resident_large.sh
#   SIZES="2048 4096 16384" OLD="GDPSO_STAGE1_RESIDENT_MAX=1024" NEW="GDPSO_NONE=1" BENCH_ARGS="--dtype bf16 --z-dim 100" bash tools/resident_large.sh
EXTRA="$BENCH_ARGS"
for n in ${SIZES:-768 1536 2048 3072 4096}; do
  for v in "${OLD:-GDPSO_STAGE1_RESIDENT=0}" "${NEW:-GDPSO_STAGE1_RESIDENT_MAX=65536}"; do
    BENCH_ARGS="--particles $n --steps 20 $EXTRA" bash tools/ab_bench.sh "$v" | sed "s/^/n=$n /"
  done
done
