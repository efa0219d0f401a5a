#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_mbchain.py -q -x 2>&1 | tail -3) > gpurun_out/i17_tests.log
cat gpurun_out/i17_tests.log
true
cat gpurun_out/i17_ab.log
TAG=p17 bash tools/r04_prof.sh > /dev/null 2>&1
grep "mb_dw_bwd" gpurun_out/p17_chron.txt | tail -6
