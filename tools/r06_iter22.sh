#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_x3.py tests/test_gpu_wino_tower.py tests/test_gpu_model.py -x -q 2>&1 | tail -6 > gpurun_out/r06_iter22_tests.log
cat gpurun_out/r06_iter22_tests.log
for i in 1 2 3; do for v in "RN_WINO_PRE=0 RN_X3_BFRAG=0" "RN_WINO_PRE=0 RN_X3_BFRAG=1"; do
  echo "$v: $(env $v timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120)"
done; done | tee gpurun_out/r06_iter22_bench.log
