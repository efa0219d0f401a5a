#!/bin/bash
# round 6, fourth GPU pass: dense 1x1 convs on the split-bf16 kernels -- parity, then cfg 3 / cfg 4 / cfg 5 (fp32) with and without
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -m gpu -s 2>&1 | grep -v Warning | tail -25 > gpurun_out/r06_i4_tests_x3.log
timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_backbones.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_i4_tests.log
for v in 0 1 0 1; do
  echo "RN_X3_CONV1X1=$v $(RN_X3_CONV1X1=$v timeout 600 python tools/bench_configs.py resnet_50 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print("cfg3", r["images_per_sec"], r["ms_per_step"])')"
  echo "RN_X3_CONV1X1=$v $(RN_X3_CONV1X1=$v timeout 600 python tools/bench_configs.py densenet_121 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print("cfg4", r["images_per_sec"], r["ms_per_step"])')"
done > gpurun_out/r06_i4_cfgs.txt 2>&1
for v in 0 1; do echo "RN_X3_CONV1X1=$v"; RN_X3_CONV1X1=$v timeout 600 python tools/bench_inference.py 2>/dev/null | tail -3 | cut -c1-300; done > gpurun_out/r06_i4_inf.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity_r6.py -x -q -m gpu -s -k conditioned 2>&1 | grep -v Warning | tail -12 > gpurun_out/r06_i4_cond.log
tail -4 gpurun_out/r06_i4_tests_x3.log; cat gpurun_out/r06_i4_tests.log gpurun_out/r06_i4_cfgs.txt gpurun_out/r06_i4_inf.txt; tail -5 gpurun_out/r06_i4_cond.log | cut -c1-900
