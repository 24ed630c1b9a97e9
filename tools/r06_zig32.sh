#!/bin/bash
mkdir -p gpurun_out/r06i
timeout 1500 python -m pytest tests/test_gpu_ziggurat.py -x -q -m gpu > gpurun_out/r06i/pytest.txt 2>&1; tail -15 gpurun_out/r06i/pytest.txt
for dt in f32 f64; do
  for g in ziggurat box-muller; do
    timeout 300 python bench.py --dtype $dt --normal-gen $g --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$dt $g', d['value'], d['roofline']['frac'], d['config']['workload'][:90])"
  done
done
