#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity_r6.py -x -q -m gpu -s -k conditioned 2>&1 | grep -v Warning | grep "cfg-3 size\|product /\|product vs\|passed\|failed\|Error" | cut -c1-1500 > gpurun_out/r06_i5_cond.log
cat gpurun_out/r06_i5_cond.log
export TMPDIR=/tmp
rm -rf gpurun_out/r06_i5_prof
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_i5_prof -o c4 -- python tools/prof_cfg.py densenet_121 640 4 > gpurun_out/r06_i5_prof4.log 2>&1
python tools/family.py $(find gpurun_out/r06_i5_prof -name "c4_kernel_trace.csv" | head -1) > gpurun_out/r06_i5_cfg4_families.txt
rm -rf gpurun_out/r06_i5_prof
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_i5_prof -o c3 -- python tools/prof_cfg.py resnet_50 800 2 > gpurun_out/r06_i5_prof3.log 2>&1
python tools/family.py $(find gpurun_out/r06_i5_prof -name "c3_kernel_trace.csv" | head -1) > gpurun_out/r06_i5_cfg3_families.txt
rm -rf gpurun_out/r06_i5_prof
head -25 gpurun_out/r06_i5_cfg4_families.txt
