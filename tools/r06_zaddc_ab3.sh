#!/bin/bash
mkdir -p gpurun_out/r06m
out=gpurun_out/r06m/ab3.txt; : > $out
run() { timeout 400 python bench.py --config $1 --dtype f64 --normal-gen ziggurat --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 [$2]', '%.5g' % d['value'], '%.4f' % d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))" >> $out; }
for rep in 1 2 3; do
run c2 ""
run c2 "--lib advancedmh.jl_amd/abvar/libmhx_zaddc0.so"
done
cat $out
