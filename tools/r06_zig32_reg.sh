#!/bin/bash
# fp32 ziggurat on the register kernels (RWMH user source, MALA): parity, then the c2_user shape in both widths and generators
mkdir -p gpurun_out/r06j
timeout 1500 python -m pytest tests/test_gpu_ziggurat.py tests/test_gpu_mala.py -x -q -m "gpu" > gpurun_out/r06j/pytest.txt 2>&1; tail -8 gpurun_out/r06j/pytest.txt
for dt in f32 f64; do
  for g in ziggurat box-muller; do
    timeout 300 python bench.py --config c2 --c2-user --dtype $dt --normal-gen $g --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e 2>gpurun_out/r06j/err_$dt_$g.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c2_user $dt $g', d['value'], d['roofline']['frac'], d['config'].get('kernel_variant'))"
  done
done
