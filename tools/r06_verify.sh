#!/bin/bash
# round-6 closing verification on the GPU box: default tier, soak tier, smoke; times into gpurun_out/r06_verify.txt
mkdir -p gpurun_out
out=gpurun_out/r06_verify.txt
: > $out
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -5 >> $out
t1=$(date +%s); echo "default tier wall: $((t1-t0)) s" >> $out
timeout 1200 python -m pytest tests -q -m "gpu and soak" -p no:cacheprovider 2>&1 | tail -5 >> $out
t2=$(date +%s); echo "soak tier wall: $((t2-t1)) s" >> $out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $out 2>&1
t3=$(date +%s); echo "smoke wall: $((t3-t2)) s" >> $out
cat $out
