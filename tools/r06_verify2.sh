#!/bin/bash
# GPU default tier with its full log kept, then the soak tier
mkdir -p gpurun_out/r06v
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r06v/default.txt 2>&1
t1=$(date +%s); echo "default tier wall: $((t1-t0)) s" >> gpurun_out/r06v/default.txt
grep -E "passed|failed|error" gpurun_out/r06v/default.txt | tail -3; tail -1 gpurun_out/r06v/default.txt
timeout 1500 python -m pytest tests -q -m "gpu and soak" -p no:cacheprovider > gpurun_out/r06v/soak.txt 2>&1
t2=$(date +%s); echo "soak tier wall: $((t2-t1)) s" >> gpurun_out/r06v/soak.txt
grep -E "passed|failed|error" gpurun_out/r06v/soak.txt | tail -3; tail -1 gpurun_out/r06v/soak.txt
