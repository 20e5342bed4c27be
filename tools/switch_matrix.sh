#!/bin/bash
# The GPU test suite under each development-switch combination (run on the GPU box from the
# repository root); writes gpurun_out/switch_matrix.txt.  Stops at the first failing run.
#   bash tools/switch_matrix.sh [first row] [last row]     (rows count from 1; a run takes ~105 s: eight rows to a call)
set -e
FIRST=${1:-1}
LAST=${2:-1000}
OUT=gpurun_out/switch_matrix.txt
[ "$FIRST" -gt 1 ] || : > $OUT
ROW=0
run() {
    ROW=$((ROW + 1))
    if [ $ROW -lt $FIRST ] || [ $ROW -gt $LAST ]; then return 0; fi
    echo "== ${*:-defaults}" >> $OUT
    env "$@" timeout -k 10 300 python3 -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/switch_run.log 2>&1 || { tail -n 30 gpurun_out/switch_run.log >> $OUT; exit 1; }
    tail -n 1 gpurun_out/switch_run.log >> $OUT
}
run GDPSO_NONE=1
run GDPSO_CONV_BORDER_CLASSES=0
run GDPSO_CONV_TILE32=0
run GDPSO_CONV_PAIR=0 GDPSO_CONV_FUSE_1X1=0
run GDPSO_GEN_LAST_MFMA=0 GDPSO_STEM_MFMA=0
run GDPSO_SPLITK_FUSED_MAX=8 GDPSO_CONV_TILE32=0
run GDPSO_CONV_SNAKE=0
run GDPSO_CONV_CHAIN=0
run GDPSO_CHAIN_TEAMS=2
run GDPSO_STAGE1_RESIDENT=0
run GDPSO_STAGE2_RESIDENT=0
run GDPSO_STAGE1_RESIDENT_MIN=1 GDPSO_STAGE2_RESIDENT_MIN=1
run GDPSO_STEM_LDS_KB=64 GDPSO_STAGE1_RESIDENT=0
run GDPSO_GEN_RESIDENT=0
run GDPSO_DIST_DEFER=0 GDPSO_STAGE1_RESIDENT_MAX=1024
run GDPSO_RESIDENT_TAIL=0 GDPSO_CHAIN_MAX_IMAGES=1000000
cat $OUT
