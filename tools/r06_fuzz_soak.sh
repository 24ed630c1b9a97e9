#!/bin/bash
# the random-configuration file under six other seed slices, every tier (MHX_SOAK=1), run-time kernels by the product's compiler
mkdir -p gpurun_out/r06s
out=gpurun_out/r06s/fuzz_soak.txt; : > $out
for s in 1 2 3 4 5 6; do
  t0=$(date +%s)
  MHX_SOAK=1 MHX_FUZZ_SEED=$s timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1 > /tmp/fz.txt
  t1=$(date +%s); echo "MHX_FUZZ_SEED=$s: $(cat /tmp/fz.txt) [wall $((t1-t0)) s]" >> $out
done
cat $out
