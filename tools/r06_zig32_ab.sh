#!/bin/bash
for w in 0 1 2 3; do
  extra="--opt NO_PREBUILT=1"; [ $w -gt 0 ] && extra="$extra --opt COOP_WAVES=$w"
  for lanes in 2 4; do
  timeout 300 python bench.py --dtype f32 --lanes $lanes --normal-gen ziggurat --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e $extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves $w lanes $lanes', d['value'], d['roofline']['frac'])"
  done
done
