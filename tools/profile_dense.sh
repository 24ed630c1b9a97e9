#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + SQ counters of tools/bench_dense.py -- the matrix-core RWMH
# kernel (default) and the vector cooperative kernel (MHX_NO_MFMA=1) on the dense Gaussian target, d = 100 / 128 / 200.
# Usage: tools/profile_dense.sh <tag>
set -u
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for DT in f32 f64; do
  for NM in 0 1; do
    OUT=$REPO/gpurun_out/${TAG}_dense_${DT}_nomfma${NM}
    mkdir -p $OUT
    export MHX_DTYPE=$DT MHX_NO_MFMA=$NM DIMS="100 128 200 512"
    python $REPO/tools/bench_dense.py > $OUT/bench.jsonl 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o bench -- python $REPO/tools/bench_dense.py > $OUT/ktrace.log 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- python $REPO/tools/bench_dense.py > $OUT/pmc_sq.log 2>&1
    python $REPO/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
    grep -i "rwmh_mfma\|rwmh_dense" $OUT/summary.txt | head -12
    find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
  done
done
