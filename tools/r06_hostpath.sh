#!/bin/bash
# round 6: the accept-compacted return path on the GPU box -- tests, host-thread scaling, the bench's e2e block
mkdir -p gpurun_out/r06a
cd tools/ubench && g++ -O3 -std=c++17 -pthread -I../../advancedmh.jl_amd/csrc -o expand_harness expand_harness.cpp ../../advancedmh.jl_amd/csrc/mhx_host_expand.cpp expand_harness_fail.cpp && \
  for t in 1 2 4 8 12 16 24 32; do ./expand_harness $t | tail -1; done > ../../gpurun_out/r06a/expand_scaling.txt 2>&1
cd ../..
nproc > gpurun_out/r06a/host.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/r06a/host.txt; lscpu | grep -i "model name\|socket\|numa\|L2\|L3" >> gpurun_out/r06a/host.txt
cat /sys/kernel/mm/transparent_hugepage/enabled >> gpurun_out/r06a/host.txt
timeout 900 python -m pytest tests/test_gpu_host_path.py -x -q -m gpu > gpurun_out/r06a/pytest_hostpath.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06a/pytest_hostpath.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err; echo "bench rc $?" >> gpurun_out/r06a/bench.err
tail -5 gpurun_out/r06a/pytest_hostpath.txt; cat gpurun_out/r06a/expand_scaling.txt
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06a/bench.json").read().strip().splitlines()[-1])
    print(json.dumps(d.get("e2e_host"), indent=1))
except Exception as e:
    print("bench parse", e)
PY
