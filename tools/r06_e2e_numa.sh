#!/bin/bash
mkdir -p gpurun_out/r06a
for f in /sys/bus/pci/devices/*/numa_node; do d=$(dirname $f); if [ "$(cat $d/vendor 2>/dev/null)" = "0x1002" ]; then echo "$d class $(cat $d/class) numa $(cat $f)"; fi; done > gpurun_out/r06a/gpu_numa.txt
run() { # label, taskset-list
  if [ -n "$2" ]; then pre="taskset -c $2"; else pre=""; fi
  $pre timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess > gpurun_out/r06a/bench_$1.json 2> gpurun_out/r06a/bench_$1.err
  python - "$1" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06a/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
e = d.get("e2e_host", {})
for k in ("save_all", "save_all_pageable", "save_all_plain"):
    b = e.get(k, {})
    print(sys.argv[1], k, b.get("value"), "expand", b.get("host_expand_GBps"), "link", b.get("link_GBps"))
PY
}
run free ""
run node0 "0-63,128-191"
run node1 "64-127,192-255"
cat gpurun_out/r06a/gpu_numa.txt
