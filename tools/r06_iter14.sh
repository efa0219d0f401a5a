#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/r06_i14_prof
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_i14_prof -o i32 -- python tools/inf32_prof.py > gpurun_out/r06_i14.log 2>&1
python tools/family.py $(find gpurun_out/r06_i14_prof -name "i32_kernel_trace.csv" | head -1) 0.5 > gpurun_out/r06_i14_cfg5_f32_families.txt
python tools/by_grid.py $(find gpurun_out/r06_i14_prof -name "i32_kernel_trace.csv" | head -1) | head -40 > gpurun_out/r06_i14_cfg5_f32_by_grid.txt
rm -rf gpurun_out/r06_i14_prof
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_i14_prof -o c3 -- python tools/prof_cfg.py resnet_50 800 2 > gpurun_out/r06_i14b.log 2>&1
python tools/family.py $(find gpurun_out/r06_i14_prof -name "c3_kernel_trace.csv" | head -1) > gpurun_out/r06_i14_cfg3_families.txt
rm -rf gpurun_out/r06_i14_prof
head -30 gpurun_out/r06_i14_cfg5_f32_families.txt; head -24 gpurun_out/r06_i14_cfg3_families.txt
