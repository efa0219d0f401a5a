#!/bin/bash
# fragment-ordered kernel operand: tests + product timings + bench A/B
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_wino_tower.py -x -q 2>&1 | tail -8 > gpurun_out/r06_iter18_tests.log
cat gpurun_out/r06_iter18_tests.log
for v in "RN_X3_BFRAG=0" "RN_X3_BFRAG=1"; do echo "== $v"; env $v python tools/x3_bench.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/r06_iter18_x3.log
for i in 1 2; do for v in "RN_X3_BFRAG=0" "RN_X3_BFRAG=1"; do
  echo "$v: $(env $v timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120)"
done; done | tee gpurun_out/r06_iter18_bench.log
