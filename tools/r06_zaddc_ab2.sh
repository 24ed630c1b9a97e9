#!/bin/bash
mkdir -p gpurun_out/r06m
out=gpurun_out/r06m/ab2.txt; : > $out
run() { timeout 400 python bench.py --config $1 --dtype f64 --normal-gen ziggurat --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 [$2]', '%.5g' % d['value'], '%.4f' % d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))" >> $out; }
for rep in 1 2; do
run c2 ""
run c2 "--tools-lib --opt NO_PREBUILT=1 --opt JIT_DEFS=MHX_ZADDC=0"
run c2 "--opt NO_PREBUILT=1"
done
run c5 ""
timeout 900 python -m pytest tests/test_gpu_ziggurat.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2 >> $out
cat $out
