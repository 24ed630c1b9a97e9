#!/bin/bash
# Where a cooperative stretch-move half-step spends its time (C3): option EMCEE_PROBE = n of the TOOLS build (libmhx_tools.so) ends the kernel after phase n
# (1 launch + arguments, 2 + draws, 3 + rows and move, 4 + factor image in LDS, 5 + A y and butterfly, 6 + accept and state update without the record, 0 = the real kernel).
# Prints us per half-step launch for each.  Usage (on the GPU box): tools/emcee_probe.sh [f64|f32]
DT=${1:-f64}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for P in 1 2 3 4 5 6 0; do
  if [ $P = 0 ]; then PO=""; else PO="--tools-lib --opt EMCEE_PROBE=$P"; fi
  python $REPO/bench.py --config c3 --dtype $DT --steps 10 --warmup 2 --no-cpu-baseline --no-second-dtype $PO 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('probe $P $DT: %.2f us per half-step launch (%.3g moves/s)' % (b['roofline']['avg_launch_ms']*1e3, b['value']))"
done
