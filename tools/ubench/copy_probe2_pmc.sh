#!/bin/bash
# PMC passes over tools/ubench/copy_probe2 (looping copy k_pair / k_phase against the one-shot copy k_oneshot); run on the GPU box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04e_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for set in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_TAG_STALL_sum TCC_REQ_sum" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o b -- $REPO/tools/ubench/copy_probe2 > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $OUT/p$i.log | tr '\n' ' ')"
done
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur=collections.defaultdict(list)
for f in glob.glob(out+"/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:60]].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k,v in sorted(acc.items()):
    print(k, "avg_ns=%.4g" % (sum(dur[k])/max(1,len(dur[k]))), {c: "%.4g" % (sum(x)/len(x)) for c,x in sorted(v.items())})
PY
find $OUT -name '*.csv' -delete; find $OUT -name '*.db' -delete
