#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_mbchain.py -q -x -s 2>&1 | tail -8) > gpurun_out/i7_tests.log
cat gpurun_out/i7_tests.log
bash tools/ab.sh 3 "RN_MB_PW_BIG=0" "RN_MB_PW_BIG_MIN_HW=16384" "RN_MB_PW_BIG_MIN_HW=65536" > gpurun_out/i7_ab.log 2>&1
cat gpurun_out/i7_ab.log
bash tools/pwbig_phase.sh
