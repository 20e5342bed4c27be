#!/bin/bash
# A/B library: recompiles conv_gemm.hip with extra defines and links it with the product's other objects
# (build the product first) into tools/libgdpso_<name>.so:
#   bash tools/build_variant.sh pf3 -DGDPSO_PF=3
# Run the benchmark on it with: python tools/bench_with_lib.py tools/libgdpso_pf3.so [bench.py arguments]
set -e
NAME=$1
shift
cd "$(dirname "$0")/../gan-discovery-pso_amd/csrc"
/opt/rocm/bin/hipcc -x hip -c conv_gemm.hip -o build/conv_gemm_$NAME.o --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@"
/opt/rocm/bin/hipcc -shared -o ../../tools/libgdpso_$NAME.so build/pack.o build/conv_gemm_$NAME.o build/nets.o build/stage_resident.o build/pso.o build/vq.o --offload-arch=gfx950
echo tools/libgdpso_$NAME.so
