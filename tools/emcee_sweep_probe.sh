# Where the one-launch-per-sweep stretch move spends its time (C3; --c3-rotated: the scalar-factor form): option EMCEE_PROBE = n of the TOOLS build (libmhx_tools.so) ends the
# kernel after phase n (2 launch + own row + draws, 3 + rows and candidates in LDS, 4 (scalar form) + barrier and y in registers,
# 5 + row products and reduction, 6 + accept and new state without the record, 7 / 8 (lane-group form) accepted rows only without /
# with the record, 0 the real kernel).  us per launch.  Usage: tools/emcee_sweep_probe.sh [bench args]
for dt in f64 f32; do for P in 2 3 4 5 6 0; do
  if [ $P = 0 ]; then PO=""; else PO="--tools-lib --opt EMCEE_PROBE=$P"; fi
  python bench.py --config c3 --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline --no-second-dtype $PO "$@" 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('probe $P $dt: %.2f us per launch (%.3g moves/s)' % (b['roofline']['avg_launch_ms']*1e3, b['value']))"
done; done
