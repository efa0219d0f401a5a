#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_mbchain.py -q -x 2>&1 | tail -3) > gpurun_out/i15_tests.log
RN_MB_DWF_TH=16 RN_MB_DWB_TH=16 timeout 900 python -m pytest tests/test_gpu_mbchain.py -q -x 2>&1 | tail -3 >> gpurun_out/i15_tests.log
cat gpurun_out/i15_tests.log
bash tools/ab.sh 3 "RN_MB_DWF_TH=8" "RN_MB_DWF_TH=16" "RN_MB_DWF_TH=16 RN_MB_DWF_TW=16" "RN_MB_DWB_TH=16" "RN_MB_DWB_TH=16 RN_MB_DWB_TW=16" "RN_MB_DWF_TH=16 RN_MB_DWB_TH=16" > gpurun_out/i15_ab.log 2>&1
cat gpurun_out/i15_ab.log
