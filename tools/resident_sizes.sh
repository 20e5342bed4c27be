for n in 8 32 64 128 192 256 384 768 1024; do
  for v in "GDPSO_STAGE1_RESIDENT=0" "GDPSO_STAGE2_RESIDENT=0" "GDPSO_STAGE2_RESIDENT=1"; do
    BENCH_ARGS="--particles $n" bash tools/ab_bench.sh "$v" | sed "s/^/n=$n /"
  done
done
