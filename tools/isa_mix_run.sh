#!/bin/bash
# Run on the GPU box (via gpurun): the PMC class-counter passes tools/isa_mix.py needs, for one bench configuration, in SEPARATE
# rocprofv3 runs (--pmc only: never combined with a trace domain).  Usage: [DTYPE=f32] tools/isa_mix_run.sh <tag> <config c2|c5> [bench args]
# (the run-time kernels of the run are kept under <out>/jit: tools/isa_mix.py --hsaco reads the mnemonic mix of a JIT kernel from there)
set -u
TAG=$1; CFG=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
DTYPE=${DTYPE:-f64}
OUT=$REPO/gpurun_out/${TAG}_${CFG}
mkdir -p $OUT/jit
export MHX_CACHE_DIR=$OUT/jit
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --config $CFG --dtype $DTYPE --steps 6 --warmup 2 --no-cpu-baseline --no-second-dtype --no-ess --no-other-configs --no-e2e $*"
cd /tmp
python $REPO/bench.py --config $CFG --dtype $DTYPE --no-cpu-baseline --no-second-dtype --no-other-configs --no-e2e --no-ess "$@" 2>/dev/null | tail -1 > $OUT/bench.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_c.log 2>&1
python $REPO/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*agent_info.csv' -delete
tail -3 $OUT/pmc_a.log
