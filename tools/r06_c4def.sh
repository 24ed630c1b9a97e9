#!/bin/bash
mkdir -p gpurun_out/r06e
timeout 900 python -m pytest tests/test_gpu_ram_deferred.py tests/test_gpu_misc.py -x -q -m gpu -k "deferred or checkpoint" > gpurun_out/r06e/pytest.txt 2>&1; tail -3 gpurun_out/r06e/pytest.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "deferred" >> gpurun_out/r06e/pytest.txt 2>&1; tail -2 gpurun_out/r06e/pytest.txt
bash tools/profile_round.sh r06e c4 f64 --c4-moving --c4-deferred 2>&1 | tail -25
