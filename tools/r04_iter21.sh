#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -q -x -k "not cfg3_cfg4_full and not twenty" 2>&1 | tail -3) > gpurun_out/i21_tests.log
cat gpurun_out/i21_tests.log
bash tools/ab.sh 3 "RN_X=0" > gpurun_out/i21_ab.log 2>&1
cat gpurun_out/i21_ab.log
