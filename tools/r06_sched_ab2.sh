#!/bin/bash
mkdir -p gpurun_out/r06t
out=gpurun_out/r06t/sched2.txt; : > $out
run() { timeout 400 python bench.py --config $1 --dtype $2 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e --tools-lib --opt NO_PREBUILT=1 ${3:+--opt "JIT_FLAGS=$3"} 2>gpurun_out/r06t/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2 [$3]', '%.5g' % d['value'], '%.4f' % d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))" >> $out || { echo "$1 $2 [$3] FAILED" >> $out; }; }
for rep in 1 2 3; do
run c2 f64 ""
run c2 f64 "-mllvm -amdgpu-sched-strategy=max-memory-clause"
run c2 f64 "-mllvm -amdgpu-use-amdgpu-trackers=1"
run c2 f64 "-mllvm -amdgpu-sched-strategy=max-memory-clause -mllvm -amdgpu-use-amdgpu-trackers=1"
done

cat $out
