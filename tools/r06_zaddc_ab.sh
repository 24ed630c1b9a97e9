#!/bin/bash
# fp64 C2 / C5: the fast path noting failures by compare + add-with-carry (MHX_ZADDC=1, hiprtc build of the tools library) against the selects
mkdir -p gpurun_out/r06m
out=gpurun_out/r06m/ab.txt; : > $out
run() { timeout 400 python bench.py --config $1 --dtype f64 --normal-gen ziggurat --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess --no-e2e $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 [$2]', '%.5g' % d['value'], '%.4f' % d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))" >> $out; }
for rep in 1 2; do
run c2 "--tools-lib --opt NO_PREBUILT=1"
run c2 "--tools-lib --opt NO_PREBUILT=1 --opt JIT_DEFS=MHX_ZADDC=1"
done
run c2 ""
timeout 600 python - >> $out 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, "advancedmh.jl_amd"); sys.path.insert(0, ".")
import mhx
from oracle import oracle as O
mhx.use_library(mhx.TOOLS_LIB_PATH)
mhx.set_option("NO_PREBUILT", "1"); mhx.set_option("JIT_DEFS", "MHX_ZADDC=1")
mhx.set_default_dtype("f64"); O.set_dtype("f64")
for d, C, N, lanes in [(100, 70, 30, 2), (98, 33, 20, 2), (52, 130, 25, 1), (200, 40, 12, 4)]:
    s = float(np.float32(2.38 / d ** 0.5))
    ch = mhx.sample(mhx.DensityModel(mhx.IsoGaussian(d)), mhx.RWMH(mhx.MvNormal(mhx.zeros(d), s * s * mhx.I)), N, C, seed=5 + d, first_chain=3, reduce_lanes=lanes, normal_gen="ziggurat", allow_tainted=True)
    L = ch.stats["reduce_lanes"]
    ref = O.rwmh(O.iso_gauss(d, reduce_lanes=L), O.Proposal(O.PROP_ISO, s, normal_gen=1), O.schedule(N), 5 + d, 3, C)
    print("parity d=%d lanes=%d variant=%d:" % (d, L, ch.stats["kernel_variant"]), np.array_equal(ch.value.view(np.uint64), ref["samples"].view(np.uint64)))
PY
cat $out
