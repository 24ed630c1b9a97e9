#!/bin/bash
mkdir -p gpurun_out/r06q
out=gpurun_out/r06q/fence.txt; : > $out
run() { timeout 400 python bench.py --config c2 --dtype f32 --normal-gen ziggurat --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e --tools-lib --opt NO_PREBUILT=1 $1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f32 c2 [$1]', '%.5g' % d['value'], '%.4f' % d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))" >> $out; }
for rep in 1 2; do
for f in 3 2 1 0; do run "--opt JIT_DEFS=MHX_ZIG32_FENCE=$f"; done
done
cat $out
