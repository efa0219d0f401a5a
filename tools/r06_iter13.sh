#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06_i13_tests.log
for v in 0 1; do echo "RN_X3_IM2COL=$v"; RN_X3_IM2COL=$v timeout 600 python tools/bench_inference.py 2>/dev/null | tail -3 | head -1 | cut -c1-200; done > gpurun_out/r06_i13_inf.txt 2>&1
timeout 600 python tools/bench_configs.py resnet_50 2>/dev/null | cut -c1-160 >> gpurun_out/r06_i13_inf.txt
cat gpurun_out/r06_i13_tests.log gpurun_out/r06_i13_inf.txt
