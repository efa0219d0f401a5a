#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_mbchain.py -q -x 2>&1 | tail -3) > gpurun_out/i17_tests.log
cat gpurun_out/i17_tests.log
bash tools/ab.sh 3 "RN_MB_DWB_PF=0" "RN_MB_DWB_PF=1" > gpurun_out/i17_ab.log 2>&1
cat gpurun_out/i17_ab.log
TAG=p17 bash tools/r04_prof.sh > /dev/null 2>&1
grep "mb_dw_bwd" gpurun_out/p17_chron.txt | tail -6
