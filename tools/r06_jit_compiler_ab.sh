#!/bin/bash
# run-time kernels built by the installation's clang++ (default) against hiprtc as the process has it (inside Python: the PyTorch wheel's)
mkdir -p gpurun_out/r06p
out=gpurun_out/r06p/ab.txt; : > $out
export MHX_CACHE_DIR=""            # every kernel compiled here and now
one() { # config, extra flags, label
  timeout 900 python bench.py --config $1 --steps ${4:-20} --warmup 3 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e $2 2>gpurun_out/r06p/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('%-12s %-40s %-8s value=%.4g frac=%s launch_ms=%s kernel=%s' % ('$3', '$2', d['config'].get('jit_compiler'), d['value'], r.get('frac'), r.get('avg_launch_ms'), d['config'].get('kernel_variant')))" >> $out || tail -3 gpurun_out/r06p/err.txt >> $out; }
for jc in "" "--opt JIT_COMPILER=hiprtc"; do
  one c2 "--opt NO_PREBUILT=1 $jc" c2_jit
  one c2 "--c2-user $jc" c2_user
  one c2 "--c2-literal $jc" c2_literal
  one c3 "$jc" c3 200
  one c3 "--c3-rotated $jc" c3_rotated 200
  one c3 "--c3-user $jc" c3_user 200
  one c3 "--c3-small $jc" c3_small 5
  one c5 "--c5-banana $jc" c5_banana 10
  one c4 "$jc" c4 3
  one c4 "--c4-deferred $jc" c4_deferred 3
  one c1 "$jc" c1 3
done
cat $out
