# C3 sweep kernel: lanes per walker x waves per block (us per launch by HIP events); usage: tools/emcee_lanes_ab.sh [extra bench args]
for dt in f64 f32; do for L in 4 8 16 32; do for WV in 4 8; do
  echo -n "$dt lanes=$L waves=$WV: "; python bench.py --opt EMCEE_WAVES=$WV --config c3 --dtype $dt --lanes $L --steps 10 --warmup 2 --no-cpu-baseline --no-second-dtype "$@" 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.2f us per launch (%.4g moves/s) %s lanes %d' % (b['roofline']['avg_launch_ms']*1e3, b['value'], b['config']['kernel_variant'], b['config']['lanes_per_unit']))"
done; done; done
