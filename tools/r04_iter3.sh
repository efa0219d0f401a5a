#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest "tests/test_gpu_fullsize.py::test_cfg5_fp16_detections_match_the_fp32_oracle" -q -x -s 2>&1 | grep -E "fp16 detections vs|passed|failed|^E " ) > gpurun_out/i3_tests_c.log
bash tools/ab.sh 3 "RN_MB_DW_BLOCKS=512" "RN_MB_DW_BLOCKS=1024" "RN_MB_DW_BLOCKS=768 RN_MB_COMPACT_ABOVE=64" "RN_MB_COMPACT_ABOVE=64" "RN_MB_COMPACT_ABOVE=16" > gpurun_out/i3_ab.log 2>&1
cat gpurun_out/i3_tests_c.log; cat gpurun_out/i3_ab.log
