#!/bin/bash
# Usage: tools/profile_sq.sh <tag> <command...>  -- SQ instruction-mix counters only (two passes)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="$*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o bench -- bash -c "cd $REPO && $CMD" > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -o bench -- bash -c "cd $REPO && $CMD" > $OUT/pmc_sq2.log 2>&1
python $REPO/tools/summarize_profile.py $OUT 2>&1 | grep -v "__amd_rocclr\|k_ram_init\|k_ram_repack\|k_ram_unit\|k_ram_diag"
