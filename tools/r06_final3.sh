#!/bin/bash
# closing pass: rocprofv3 of the stretch-move configurations on the kernels as the product builds them now (clang++), then the driver-shaped line
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/profile_round.sh r06y c3 f64 2>&1 | tail -4
bash tools/profile_round.sh r06ysml c3 f64 --c3-small 2>&1 | tail -4
bash tools/profile_round.sh r06yrot c3 f64 --c3-rotated 2>&1 | tail -4
rm -rf ~/.cache/mhx          # the line as a fresh box produces it: every run-time kernel compiled inside the run
t0=$(date +%s)
bash tools/r06_bench_full.sh
t1=$(date +%s); echo "full bench wall (cold kernel cache): $((t1-t0)) s"
