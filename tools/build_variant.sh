#!/bin/bash
# a variant of libmhx.so whose fp64 engine is compiled with extra -D flags (A/B experiments on tuning macros):
#   tools/build_variant.sh name -DMHX_RAM_DEFER_K=4 ...   ->  advancedmh.jl_amd/abvar/libmhx_<name>.so  (git-ignored, travels with gpurun; delete after the A/B)
set -e
name=$1; shift
cd "$(dirname "$0")/../advancedmh.jl_amd/csrc"
make -s all
mkdir -p ../abvar
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-pass-failed -Wno-array-bounds"
hipcc $FLAGS -DMHX_REAL64=1 "$@" -c -o ../abvar/mhx_api_f64_$name.o mhx_api.hip
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../abvar/libmhx_$name.so mhx_api_f32.o ../abvar/mhx_api_f64_$name.o mhx_abi.o mhx_comm.o mhx_group.o mhx_host_expand.o mhx_jit_ext.o -lhiprtc -ldl -lpthread
rm -f ../abvar/mhx_api_f64_$name.o
echo built ../abvar/libmhx_$name.so
