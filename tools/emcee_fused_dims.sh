# one launch per sweep against two at the size threshold, other dimensions (the rule was measured at d = 50)
for D in ${DIMS:-10 24 64}; do for W in ${SIZES:-16384 24576}; do for rot in "" "--c3-rotated"; do for F in 0 1; do
  echo -n "d=$D W=$W $rot fused=$F f64: "; python bench.py --opt EMCEE_FUSED=$F --config c3 --dtype f64 --dim $D --chains $W --inner 200 --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype $rot 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.2f us per launch (%.4g moves/s) %s lanes %d' % (b['roofline']['avg_launch_ms']*1e3, b['value'], b['config']['kernel_variant'], b['config']['lanes_per_unit']))"
done; done; done; done
