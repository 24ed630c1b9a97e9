#!/bin/bash
# quick PMC comparison of one bench command: two counter passes, per-kernel means printed.  Usage: tools/pmc_quick.sh <outdir> <bench args...>
set -u
OUT=$(realpath -m $1); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-dtype --no-ess --no-other-configs --no-e2e $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $OUT/p1 -o b -- $BENCH > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/p2 -o b -- $BENCH > $OUT/p2.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for p in ("p1","p2"):
    for f in glob.glob(out+"/"+p+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            if "coop" in k or "ram<" in k or "emcee" in k:
                print(p, k, {c: "%.4g" % (sum(x)/len(x)) for c,x in v.items()}, "n=%d" % len(next(iter(v.values()))))
PY
find $OUT -name '*.csv' -delete
