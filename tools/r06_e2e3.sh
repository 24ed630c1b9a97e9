#!/bin/bash
mkdir -p gpurun_out/r06c
for i in 1 2 3; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess $1 > gpurun_out/r06c/bench_$i.json 2> gpurun_out/r06c/bench_$i.err
python - $i <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06c/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
e = d.get("e2e_host", {})
for k in ("save_all", "save_all_pageable", "save_all_fresh", "save_all_plain"):
    b = e.get(k, {})
    print(sys.argv[1], k, b.get("value"), "expand", b.get("host_expand_GBps"), "link", b.get("link_GBps"))
PY
done
