#!/bin/bash
mkdir -p gpurun_out/r06r
timeout 1500 python -m pytest tests/test_gpu_ziggurat.py tests/test_gpu_fullsize.py tests/test_gpu_mala.py -x -q -m gpu > gpurun_out/r06r/pytest.txt 2>&1; tail -4 gpurun_out/r06r/pytest.txt
for rep in 1 2; do
for cfg in "c2" "c5" "c2 --c2-user"; do
timeout 400 python bench.py --config $cfg --dtype f64 --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', '%.5g' % d['value'], '%.4f' % d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))"
done; done
