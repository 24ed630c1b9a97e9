for W in ${SIZES:-1024 4096 65536 262144}; do for rot in "" "--c3-rotated"; do for F in 0 1; do for dt in f64 f32; do
  echo -n "W=$W $rot fused=$F $dt: "; python bench.py --opt EMCEE_FUSED=$F --config c3 --dtype $dt --chains $W --inner 200 --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype $rot 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.2f us per launch (%.4g moves/s) %s lanes %d' % (b['roofline']['avg_launch_ms']*1e3, b['value'], b['config']['kernel_variant'], b['config']['lanes_per_unit']))"
done; done; done; done
