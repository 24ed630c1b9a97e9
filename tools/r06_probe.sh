#!/bin/bash
for p in 0 1; do for dt in f64 f32; do
timeout 400 python bench.py --config c2 --dtype $dt --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e --tools-lib --opt NO_PREBUILT=1 --opt ZIG_PROBE=$p 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('probe=$p $dt', '%.5g' % d['value'], d['roofline'].get('avg_launch_ms'))"
done; done
