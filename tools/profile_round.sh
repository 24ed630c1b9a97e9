#!/bin/bash
# Run on the GPU box (via gpurun): for one bench configuration -- the unprofiled bench line, a rocprofv3 kernel trace of
# the same command, and SEPARATE --pmc passes for FETCH_SIZE, WRITE_SIZE and the SQ counters (never combined with trace
# domains).  Usage: tools/profile_round.sh <tag> <config c2|c3|c4|c5> <dtype f64|f32> [extra bench args...]
set -u
TAG=$1; CFG=$2; DT=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${TAG}_${CFG}_${DT}
mkdir -p $OUT
export TMPDIR=/tmp
# bench.py spins the GPU up with 30 untimed launches before the warm-up, so the traced launches run at the steady clock
BENCH="python $REPO/bench.py --config $CFG --dtype $DT --steps 10 --warmup 2 --no-cpu-baseline --no-second-dtype --no-ess --no-other-configs --no-e2e $*"
cd /tmp
echo "== bench (unprofiled)"; python $REPO/bench.py --config $CFG --dtype $DT --no-cpu-baseline --no-second-dtype --no-other-configs --no-e2e "$@" 2>/dev/null | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json
# the trace pass runs 100 timed launches so that its average is the steady-state one (the ~20 launches after idle run below the steady clock)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o bench -- ${BENCH/--steps 10/--steps 100} > $OUT/ktrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
python $REPO/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
grep -v "__amd_rocclr\|k_ram_init\|k_ram_repack\|k_ram_unit\|k_ram_diag\|k_diag\|rocprim\|k_record\|k_sum" $OUT/summary.txt | head -40
# keep the merge small: the raw per-dispatch CSVs stay on the box
find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
