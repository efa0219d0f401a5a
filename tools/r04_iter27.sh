#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_mbchain.py tests/test_gpu_fullsize.py -q -x -k "not cfg3_cfg4_full and not twenty and not fp16 and not f16" 2>&1 | tail -3) > gpurun_out/i27_tests.log
cat gpurun_out/i27_tests.log
bash tools/ab.sh 3 "RN_STEM_DIRECT=0" "RN_STEM_DIRECT=1" > gpurun_out/i27_ab.log 2>&1
cat gpurun_out/i27_ab.log
TAG=p27 bash tools/r04_prof.sh > /dev/null 2>&1
head -8 gpurun_out/p27_chron.txt
