# A/B of the ziggurat fix-up group size KS (libraries built with -DMHX_ZIG_KS_FORCE=k: libmhx_ks1.so, libmhx_ks4.so; default build: the fit rule)
for rep in 1 2; do for lib in libmhx_ks1.so libmhx.so libmhx_ks4.so; do for cfg in "c5" "c5 --c5-banana" "c2"; do
  echo -n "$lib $cfg: "; MHX_LIB=$PWD/advancedmh.jl_amd/$lib python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-second-dtype --no-ess --no-other-configs --no-e2e 2>/dev/null | tail -1 |
    python -c "import json,sys; b=json.loads(sys.stdin.read()); print('%.4g steps/s  %.3f ms per launch  %s' % (b['value'], b['roofline']['avg_launch_ms'], b['config']['kernel_variant']))"
done; done; done
