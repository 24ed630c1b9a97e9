mkdir -p gpurun_out/r06a
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-second-dtype --no-ess > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err; echo "bench rc $?" >> gpurun_out/r06a/bench.err
tail -3 gpurun_out/r06a/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06a/bench.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("e2e_host"), indent=1))
PY
