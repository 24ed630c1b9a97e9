#!/bin/bash
# code-generation flags on the cooperative RWMH kernel (run-time build by clang++, tools library): scheduler strategies
mkdir -p gpurun_out/r06t
out=gpurun_out/r06t/sched.txt; : > $out
run() { timeout 400 python bench.py --config $1 --dtype $2 --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e --tools-lib --opt NO_PREBUILT=1 ${3:+--opt "JIT_FLAGS=$3"} 2>gpurun_out/r06t/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $2 [$3]', '%.5g' % d['value'], '%.4f' % d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))" >> $out || { echo "$1 $2 [$3] FAILED: $(tail -2 gpurun_out/r06t/err.txt | cut -c1-200)" >> $out; }; }
for dt in f64 f32; do
run c2 $dt ""
run c2 $dt "-mllvm -amdgpu-sched-strategy=max-ilp"
run c2 $dt "-mllvm -amdgpu-sched-strategy=max-memory-clause"
run c2 $dt "-mllvm -amdgpu-sched-strategy=iterative-ilp"
run c2 $dt "-mllvm -amdgpu-sched-strategy=iterative-minreg"
run c2 $dt "-mllvm -amdgpu-schedule-relaxed-occupancy=true"
run c2 $dt "-mllvm -amdgpu-use-amdgpu-trackers=1"
run c2 $dt "-O2"
run c2 $dt ""
done
run c5 f64 ""
run c5 f64 "-mllvm -amdgpu-sched-strategy=max-ilp"
run c5 f64 "-mllvm -amdgpu-sched-strategy=iterative-ilp"
cat $out
