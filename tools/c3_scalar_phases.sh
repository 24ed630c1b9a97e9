# phases of the scalar-factor stretch move (MHX_EMCEE_PROBE) for a few shapes; run on the GPU box
for cfg in "8 32 1" "8 64 1" "8 32 0" "4 32 1" "16 64 1"; do set -- $cfg; for p in 3 5 6 0; do
  if [ $p = 0 ]; then unset MHX_EMCEE_PROBE; else export MHX_EMCEE_PROBE=$p; fi
  echo -n "waves=$1 wpb=$2 mode=$3 probe=$p: "; MHX_EMCEE_SCAL_MODE=$3 MHX_EMCEE_SCAL_WPB=$2 MHX_EMCEE_SCALAR=$1 python tools/c3_scalar_probe.py ${DT:-f64} time 50 16384 2>&1 | grep "^time" | sed 's/.*lanes [0-9]*: //'
done; done
for dt in f64 f32; do for cfg in "8 32" "8 64" "4 16" "16 64"; do set -- $cfg; for shape in "50 200" "33 64" "96 256" "17 70"; do MHX_EMCEE_SCAL_WPB=$2 MHX_EMCEE_SCALAR=$1 python tools/c3_scalar_probe.py $dt parity $shape 2>&1 | grep parity; done; done; done
