#!/bin/bash
mkdir -p gpurun_out/r06b
timeout 1500 python bench.py > gpurun_out/r06b/bench_full.json 2> gpurun_out/r06b/bench_full.err; echo "rc $?" >> gpurun_out/r06b/bench_full.err
tail -3 gpurun_out/r06b/bench_full.err
python - <<'PY'
import json
line = open("gpurun_out/r06b/bench_full.json").read().strip().splitlines()[-1]
print("line length", len(line))
d = json.loads(line)
print("value", d["value"], "frac", d["roofline"]["frac"])
print(json.dumps(d.get("e2e_host"), indent=0)[:1500])
for k, v in d.get("configs", {}).items():
    print(k, json.dumps({a: b for a, b in v.items() if a != "cpu"}))
PY
