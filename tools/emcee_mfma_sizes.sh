for W in ${SIZES:-4096 16384 65536 262144}; do for M in 0 1; do for dt in f64 f32; do
  echo -n "W=$W mfma=$M $dt: "; python bench.py --opt EMCEE_MFMA=$M --config c3 --c3-rotated --dtype $dt --chains $W --inner 200 --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.2f us per launch (%.4g moves/s) %s lanes %d, %d launches/step' % (b['roofline']['avg_launch_ms']*1e3, b['value'], b['config']['kernel_variant'], b['config']['lanes_per_unit'], b['config']['launches_per_step']))"
done; done; done
