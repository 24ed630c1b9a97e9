# interleaved A/B of library builds on bench configs: tools/lib_ab.sh "libA.so libB.so" "c2|c5|c5 --c5-banana ..." [reps]
LIBS=$1; CFGS=$2; REPS=${3:-2}
IFS='|' read -ra CF <<< "$CFGS"
for rep in $(seq $REPS); do for lib in $LIBS; do for cfg in "${CF[@]}"; do
  echo -n "$lib $cfg: "; MHX_LIB=$PWD/advancedmh.jl_amd/$lib python bench.py  --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-second-dtype --no-ess --no-other-configs --no-e2e 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.4g steps/s  %.3f ms per launch  %s' % (b['value'], b['roofline']['avg_launch_ms'], b['config']['kernel_variant']))"
done; done; done
