#!/bin/bash
mkdir -p gpurun_out/r06m
out=gpurun_out/r06m/c5.txt; : > $out
run() { timeout 400 python bench.py --config c5 --dtype f64 --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e --tools-lib --opt NO_PREBUILT=1 $1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c5 [$1]', '%.5g' % d['value'], d['roofline'].get('avg_launch_ms'), d['config'].get('kernel_variant'))" >> $out; }
for rep in 1 2; do
run "--opt JIT_DEFS=MHX_ZADDC=1"
run "--opt JIT_DEFS=MHX_ZADDC=2"
run "--opt JIT_DEFS=MHX_ZADDC=0"
done
cat $out
