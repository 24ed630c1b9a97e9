#!/bin/bash
# GPU default tier + soak tier with the offline compiler's failures shown (MHX_JIT_VERBOSE), the kernel cache off
mkdir -p gpurun_out/r06v
export MHX_JIT_VERBOSE=1
t0=$(date +%s)
timeout 1700 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r06v/default.txt 2>&1
t1=$(date +%s); echo "default tier wall: $((t1-t0)) s" >> gpurun_out/r06v/default.txt
grep -E "passed|failed|error" gpurun_out/r06v/default.txt | tail -3; tail -1 gpurun_out/r06v/default.txt
grep -c "offline compiler failed" gpurun_out/r06v/default.txt; grep "offline compiler failed" gpurun_out/r06v/default.txt | cut -c1-300 | sort | uniq -c | head -20
timeout 1500 python -m pytest tests -q -m "gpu and soak" -p no:cacheprovider > gpurun_out/r06v/soak.txt 2>&1
t2=$(date +%s); echo "soak tier wall: $((t2-t1)) s" >> gpurun_out/r06v/soak.txt
grep -E "passed|failed|error" gpurun_out/r06v/soak.txt | tail -3; tail -1 gpurun_out/r06v/soak.txt
grep "offline compiler failed" gpurun_out/r06v/soak.txt | cut -c1-300 | sort | uniq -c | head
