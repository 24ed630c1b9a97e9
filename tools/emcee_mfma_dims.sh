# the matrix-core form against the scalar-factor form at other dimensions (the default rule was measured at d = 50)
for D in ${DIMS:-10 24 32}; do for W in ${SIZES:-2048 4096 8192 16384}; do for M in 0 1; do
  echo -n "d=$D W=$W mfma=$M ${DT:-f64}: "; python bench.py --opt EMCEE_MFMA=$M --config c3 --c3-rotated --dtype ${DT:-f64} --dim $D --chains $W --inner 200 --steps 5 --warmup 2 --no-cpu-baseline --no-second-dtype 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.2f us per launch (%.4g moves/s) %s lanes %d, %d launches/step' % (b['roofline']['avg_launch_ms']*1e3, b['value'], b['config']['kernel_variant'], b['config']['lanes_per_unit'], b['config']['launches_per_step']))"
done; done; done
