#!/bin/bash
# where does the host side of the compacted path stop scaling?  (host only)
mkdir -p gpurun_out/r06a
cd tools/ubench && g++ -O3 -std=c++17 -pthread -I../../advancedmh.jl_amd/csrc -o expand_harness expand_harness.cpp ../../advancedmh.jl_amd/csrc/mhx_host_expand.cpp expand_harness_fail.cpp || exit 1
o=../../gpurun_out/r06a/expand_numa.txt
{
which taskset numactl
echo "== unpinned T=16, 16 blocks (4.2 GB), pre-touched 4K pages"; ./expand_harness 16 16 0 0 | tail -2
echo "== unpinned T=16, THP advised, pre-touched"; ./expand_harness 16 16 1 0 | tail -2
echo "== unpinned T=16, THP advised, first touch inside"; ./expand_harness 16 16 1 1
echo "== unpinned T=16, 4K pages, first touch inside"; ./expand_harness 16 16 0 1
echo "== taskset node0 cores 0-15, THP"; taskset -c 0-15 ./expand_harness 16 16 1 0 | tail -2
echo "== taskset node0 cores 0-15, 4K"; taskset -c 0-15 ./expand_harness 16 16 0 0 | tail -2
echo "== taskset node1 cores 64-79, THP"; taskset -c 64-79 ./expand_harness 16 16 1 0 | tail -2
echo "== taskset split 0-7,64-71, THP"; taskset -c 0-7,64-71 ./expand_harness 16 16 1 0 | tail -2
echo "== taskset node0 0-31 T=32 THP"; taskset -c 0-31 ./expand_harness 32 16 1 0 | tail -2
echo "== taskset node0 0-15 T=8 THP"; taskset -c 0-15 ./expand_harness 8 16 1 0 | tail -2
echo "== SMT pairs 0-7,128-135 T=16 THP"; taskset -c 0-7,128-135 ./expand_harness 16 16 1 0 | tail -2
} > $o 2>&1
cat $o
