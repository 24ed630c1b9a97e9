# RWMH register kernel with a user log-density, fp64: state coordinates kept in registers (MHX_REG_XR, the rest in LDS) x the private-array
# unroll threshold (MHX_REG_UNROLL) -- steps/s at 65 536 chains.  usage: tools/reg_xr_sweep.sh "d:xr,xr,... d:xr,..." "ut ut"
for spec in ${1:-100:0,30,60,85,100 128:0,30,57}; do d=${spec%%:*}; for UT in ${2:-0 100000}; do for XR in $(echo ${spec#*:} | tr , ' '); do
  echo -n "d=$d ut=$UT xr=$XR: "; MHX_REG_UNROLL=$UT MHX_REG_XR=$XR DIMS=$d MHX_DTYPE=${DT:-f64} python tools/bench_rwmh_user.py | python -c "import json,sys; print('%.3g' % json.loads(sys.stdin.read())['steps_per_s'])"
done; done; done
