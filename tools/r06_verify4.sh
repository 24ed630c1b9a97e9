#!/bin/bash
# closing check of the final tree as the driver runs it: default GPU tier, smoke, the default bench line (cold kernel cache)
mkdir -p gpurun_out/r06w
t0=$(date +%s)
timeout 1700 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r06w/default.txt 2>&1
t1=$(date +%s); echo "default tier wall: $((t1-t0)) s" >> gpurun_out/r06w/default.txt
grep -E "passed|failed|error" gpurun_out/r06w/default.txt | tail -2; tail -1 gpurun_out/r06w/default.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
rm -rf ~/.cache/mhx
t2=$(date +%s)
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06w/bench.json 2> gpurun_out/r06w/bench.err; echo "bench rc $?"
t3=$(date +%s); echo "bench wall: $((t3-t2)) s"
tail -1 gpurun_out/r06w/bench.json | python3 -c "
import json,sys
l=sys.stdin.read().strip(); d=json.loads(l); print(len(l), d['value'], d['roofline']['frac'], d['f32']['frac'], d['e2e_host']['save_all']['value'], d['config'].get('jit'), d['cpu_baseline']['value'])"
