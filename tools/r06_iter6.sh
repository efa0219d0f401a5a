#!/bin/bash
# round 6: the direct grouped 3x3 kernels -- parity, then cfg 3 with and without
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_backbones.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_i6_tests.log
for v in 0 1 0 1; do
  echo "RN_GCONV_DIRECT=$v $(RN_GCONV_DIRECT=$v timeout 600 python tools/bench_configs.py resnet_50 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print("cfg3", r["images_per_sec"], r["ms_per_step"])')"
done > gpurun_out/r06_i6_cfgs.txt 2>&1
for v in 0 1; do echo "RN_GCONV_DIRECT=$v"; RN_GCONV_DIRECT=$v timeout 600 python tools/bench_inference.py 2>/dev/null | tail -3 | head -1 | cut -c1-200; done > gpurun_out/r06_i6_inf.txt 2>&1
rm -rf gpurun_out/r06_i6_prof
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_i6_prof -o c3 -- python tools/prof_cfg.py resnet_50 800 2 > gpurun_out/r06_i6_prof3.log 2>&1
python tools/family.py $(find gpurun_out/r06_i6_prof -name "c3_kernel_trace.csv" | head -1) > gpurun_out/r06_i6_cfg3_families.txt
python tools/by_grid.py $(find gpurun_out/r06_i6_prof -name "c3_kernel_trace.csv" | head -1) | grep "gconv\|conv_fwd\|conv_dgrad\|conv_wgrad" > gpurun_out/r06_i6_cfg3_by_grid.txt
rm -rf gpurun_out/r06_i6_prof
cat gpurun_out/r06_i6_tests.log gpurun_out/r06_i6_cfgs.txt gpurun_out/r06_i6_inf.txt; head -12 gpurun_out/r06_i6_cfg3_families.txt; cat gpurun_out/r06_i6_cfg3_by_grid.txt | head -30
