#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_f16.py tests/test_gpu_fullsize.py -q -x -k "f16 or fp16" 2>&1 | tail -4) > gpurun_out/i23_tests.log
cat gpurun_out/i23_tests.log
for v in 0 1 0 1; do echo "RN_F16_HEAD_STATS=$v: $(RN_F16_HEAD_STATS=$v timeout 300 python tools/bench_inference.py 2>&1 | grep f16 | cut -c1-200)"; done > gpurun_out/i23_inf.log 2>&1
cat gpurun_out/i23_inf.log
