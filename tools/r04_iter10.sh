#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_gpu_train_cli.py tests/test_gpu_mbchain.py -q -x 2>&1 | tail -6) > gpurun_out/i10_tests.log
cat gpurun_out/i10_tests.log
bash tools/ab.sh 3 "RN_FLUSH_MIDWAY=0 RN_FUSED_OPT_NORM=0" "RN_FLUSH_MIDWAY=1 RN_FUSED_OPT_NORM=0" "RN_FLUSH_MIDWAY=1 RN_EARLY_HEAD_UPDATE=0" "RN_FLUSH_MIDWAY=1" > gpurun_out/i10_ab.log 2>&1
cat gpurun_out/i10_ab.log
