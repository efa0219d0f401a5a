#!/bin/bash
# round 6: strided dense convs through the patch matrix -- parity, cfg 3 / cfg 5 fp32 with and without, the headline once
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_x3.py tests/test_gpu_backbones.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_i7_tests.log
for v in 0 1 0 1; do
  echo "RN_X3_IM2COL=$v $(RN_X3_IM2COL=$v timeout 600 python tools/bench_configs.py resnet_50 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print("cfg3", r["images_per_sec"], r["ms_per_step"], r["peak_mem_GB"])')"
done > gpurun_out/r06_i7_cfgs.txt 2>&1
for v in 0 1; do echo "RN_X3_IM2COL=$v"; RN_X3_IM2COL=$v timeout 600 python tools/bench_inference.py 2>/dev/null | tail -3 | head -1 | cut -c1-200; done > gpurun_out/r06_i7_inf.txt 2>&1
timeout 600 python tools/bench_configs.py densenet_121 2>/dev/null | cut -c1-200 >> gpurun_out/r06_i7_cfgs.txt
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-nms --no-roofline 2>/dev/null | grep '^{' | cut -c1-200 >> gpurun_out/r06_i7_cfgs.txt
cat gpurun_out/r06_i7_tests.log gpurun_out/r06_i7_cfgs.txt gpurun_out/r06_i7_inf.txt
