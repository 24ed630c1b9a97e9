# interleaved A/B of library builds on the C3 bench line (us per launch by HIP events): tools/c3_lib_ab.sh "libA.so libB.so" [reps] [extra bench args]
LIBS=$1; REPS=${2:-3}; shift 2
for rep in $(seq $REPS); do for lib in $LIBS; do for dt in f64 f32; do
  echo -n "$lib $dt: "; MHX_LIB=$PWD/advancedmh.jl_amd/$lib python bench.py  --config c3 --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline --no-second-dtype "$@" 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.2f us per launch (%.4g moves/s) %s lanes %d' % (b['roofline']['avg_launch_ms']*1e3, b['value'], b['config']['kernel_variant'], b['config']['lanes_per_unit']))"
done; done; done
