#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace stats + HBM PMC passes.
# Usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline $*"
cd /tmp
echo "== bench (unprofiled)"; python $REPO/bench.py "$@" 2>&1 | tail -3 | tee $OUT/bench.json
echo "== kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o bench -- $BENCH > $OUT/ktrace.log 2>&1
echo "== pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "== pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
echo "== pmc SQ"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
find $OUT -name '*.csv' | head -20
python $REPO/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
