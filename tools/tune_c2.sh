#!/bin/bash
# C2 lanes-per-chain x waves-per-SIMD sweep (run on the GPU box): one bench line per shape, fp64 and fp32
out=${1:-gpurun_out/tune_c2.log}
: > $out
for dt in f64 f32; do
  if [ $dt = f64 ]; then shapes="4:1 4:2 8:2 8:3 8:4 16:2 16:4 16:6 32:4 32:8"; else shapes="2:2 2:3 4:2 4:3 4:4 8:4 8:6 8:8 16:8"; fi
  for sh in $shapes; do
    L=${sh%%:*}; Wv=${sh##*:}
    line=$(python bench.py --opt COOP_WAVES=$Wv --dtype $dt --lanes $L --steps 20 --no-cpu-baseline --no-second-dtype --no-ess 2>/dev/null | tail -1)
    echo "$dt L=$L waves=$Wv $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("steps/s=%.4g ms/step=%.3f kernel=%s lanes=%s" % (d["value"], d["ms_per_step"], d["config"]["kernel_variant"], d["config"]["lanes_per_unit"]))' 2>&1)" | tee -a $out
  done
done
