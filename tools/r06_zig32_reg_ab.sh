#!/bin/bash
# c2_user in fp32 with the ziggurat on the register kernel: hand-back walk (REG_ZSLAB=0) against the LDS slab (1), state coordinates in registers (REG_XR)
mkdir -p gpurun_out/r06k
out=gpurun_out/r06k/ab.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_ziggurat.py -x -q -m gpu -k "register or where" > gpurun_out/r06k/pytest.txt 2>&1; tail -3 gpurun_out/r06k/pytest.txt >> $out
run() { timeout 300 python bench.py --config c2 --c2-user --dtype $1 --normal-gen $2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e $3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c2_user $1 $2 [$3]', '%.4g' % d['value'], '%.3f' % d['roofline']['frac'])" >> $out; }
run f32 box-muller ""
run f32 ziggurat "--opt REG_ZSLAB=0"
for xr in 100 64; do for w in 1 2; do run f32 ziggurat "--opt REG_ZSLAB=1 --opt REG_XR=$xr --opt REG_WAVES=$w"; done; done
run f32 ziggurat ""
cat $out
