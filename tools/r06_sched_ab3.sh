#!/bin/bash
mkdir -p gpurun_out/r06t
out=gpurun_out/r06t/sched3.txt; : > $out
run() { timeout 600 python bench.py --config $1 --dtype f64 --steps $3 --warmup 3 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e $2 $4 2>gpurun_out/r06t/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $4 [$2]', '%.5g' % d['value'], '%.4f' % d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))" >> $out || { echo "$1 [$2] FAILED" >> $out; }; }
V="--lib advancedmh.jl_amd/abvar/libmhx_mmc.so"
for rep in 1 2; do
for v in "" "$V"; do
run c2 "$v" 40
run c5 "$v" 10
run c4 "$v" 3
run c4 "$v" 3 --c4-deferred
run c4 "$v" 3 --c4-fixed
run c1 "$v" 3
done; done
cat $out
