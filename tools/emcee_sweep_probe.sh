for dt in f64 f32; do for P in 5 6 7 8 0; do
  if [ $P = 0 ]; then unset MHX_EMCEE_PROBE; else export MHX_EMCEE_PROBE=$P; fi
  python bench.py --config c3 --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline --no-second-dtype 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('probe $P $dt: %.2f us per launch (%.3g moves/s)' % (b['roofline']['avg_launch_ms']*1e3, b['value']))"
done; done
