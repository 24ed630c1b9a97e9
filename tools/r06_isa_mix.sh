#!/bin/bash
# class counters for the round's final kernels: C2 in both widths, C5, c5_banana (a run-time kernel)
bash tools/isa_mix_run.sh r06_isa c2 2>&1 | tail -2
DTYPE=f32 bash tools/isa_mix_run.sh r06_isa32 c2 2>&1 | tail -2
bash tools/isa_mix_run.sh r06_isaban c5 --c5-banana 2>&1 | tail -2
ls gpurun_out/r06_isaban_c5/jit | head
