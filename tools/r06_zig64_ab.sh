#!/bin/bash
mkdir -p gpurun_out/r06r
out=gpurun_out/r06r/ab.txt; : > $out
run() { timeout 400 python bench.py --config $1 --dtype f64 --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e --tools-lib --opt NO_PREBUILT=1 $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 [$2]', '%.5g' % d['value'], '%.4f' % d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))" >> $out; }
for rep in 1 2; do
for cfg in c2 c5; do
run $cfg "--opt JIT_DEFS=MHX_ZIG64_OLD=1"
run $cfg "--opt JIT_DEFS=MHX_ZIG64_SIGNED_OK=0"
run $cfg ""
done; done
cat $out
