#!/bin/bash
# A/B runs of bench.py under environment switches: bash tools/ab_bench.sh "NAME=VAL ..." "NAME=VAL ..." ...
# One line per variant: particle-evals/s, ms per step, roofline fraction, launches, conv ms per step.
# A variant may name another build of the library (tools/build_variant.sh): "LIB=tools/libgdpso_x.so NAME=VAL".
for v in "$@"; do
  prog="bench.py"
  for tok in $v; do case "$tok" in LIB=*) prog="tools/bench_with_lib.py ${tok#LIB=}";; esac; done
  out=$(env $v timeout -k 10 150 python $prog --steps 50 --warmup 5 --no-cpu-baseline --no-reference-shape $BENCH_ARGS 2>/dev/null)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-60s %10.1f %8.4f ms  frac %.4f  launches %d  conv %.4f ms  exec %.4f' % ('$v', d['value'], d['ms_per_step'], r['frac'], r['launches_per_step'], r['conv_ms_per_step'], r['executed_frac']))" || { echo "$v failed: stopping"; exit 1; }
done
