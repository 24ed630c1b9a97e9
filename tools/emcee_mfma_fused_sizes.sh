for W in 12288 16384 20480 24576; do for F in 0 1; do
  echo -n "W=$W mfma fused=$F f64: "; python bench.py --opt EMCEE_MFMA=1 --opt EMCEE_FUSED=$F --config c3 --c3-rotated --dtype f64 --chains $W --inner 200 --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.2f us per launch (%.4g moves/s) %s, %d launches/step' % (b['roofline']['avg_launch_ms']*1e3, b['value'], b['config']['kernel_variant'], b['config']['launches_per_step']))"
done; done
