# interleaved A/B of one env knob on the C3 bench line (us per half-step by HIP events); usage: tools/c3_ab.sh KNOB "v1 v2 ..." [reps] [extra bench args]
KNOB=$1; VALS=$2; REPS=${3:-3}; shift 3
for rep in $(seq $REPS); do for v in $VALS; do for dt in f64 f32; do
  echo -n "$KNOB=$v $dt: "; python bench.py --opt ${KNOB#MHX_}=$v --config c3 --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline --no-second-dtype "$@" 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.2f us per half-step (%.4g moves/s) %s lanes %d' % (b['roofline']['avg_launch_ms']*1e3, b['value'], b['config']['kernel_variant'], b['config']['lanes_per_unit']))"
done; done; done
