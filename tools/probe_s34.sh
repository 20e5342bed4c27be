# Stage-3/4 bottleneck shapes of ResNet50 at 512 particles: the planner's choice against
# 64x64 / 32x64 tiles with K split over workgroups (separate and in-kernel reduction).
set -e
for shape in "512 2 1024 256 1 1 0" "512 2 256 1024 1 1 0" "512 1 2048 512 1 1 0" "512 1 512 2048 1 1 0"; do
  for ts in "0 0" "64 2" "64 4" "32 2" "32 4" "128 4" "128 8"; do
    echo "== $shape tile/splits $ts"
    timeout -k 5 60 python tools/probe_one.py $shape $ts 1 200 2>/dev/null
    GDPSO_SPLITK_FUSED_MAX=16 timeout -k 5 60 python tools/probe_one.py $shape $ts 1 200 2>/dev/null
  done
done
