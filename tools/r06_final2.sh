#!/bin/bash
# closing pass of the round: rocprofv3 of C2 in both widths (kernel trace + PMC passes), then the driver-shaped line, timed
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/profile_round.sh r06y c2 f64 2>&1 | tail -6
bash tools/profile_round.sh r06y c2 f32 2>&1 | tail -6
t0=$(date +%s)
bash tools/r06_bench_full.sh
t1=$(date +%s); echo "full bench wall: $((t1-t0)) s"
