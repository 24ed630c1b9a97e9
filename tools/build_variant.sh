#!/bin/bash
# a variant of libmhx.so whose fp64 engine is compiled with extra -D flags (A/B experiments on tuning macros):
#   tools/build_variant.sh name -DMHX_RAM_DEFER_K=4 ...   ->  advancedmh.jl_amd/variants/libmhx_<name>.so  (git-ignored, travels with gpurun)
set -e
name=$1; shift
cd "$(dirname "$0")/../advancedmh.jl_amd/csrc"
make -s all
mkdir -p ../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-pass-failed -Wno-array-bounds"
hipcc $FLAGS -DMHX_REAL64=1 "$@" -c -o ../variants/mhx_api_f64_$name.o mhx_api.hip
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../variants/libmhx_$name.so mhx_api_f32.o ../variants/mhx_api_f64_$name.o mhx_abi.o mhx_comm.o mhx_group.o -lhiprtc -ldl -lpthread
rm -f ../variants/mhx_api_f64_$name.o
echo built ../variants/libmhx_$name.so
