#!/bin/bash
# Phase timing of the rn_mb_* kernels inside the real step: the bench under rocprofv3 with RN_MB_DBG=<kernel>:<phase> (the kernel returns after that
# phase; results are invalid, durations are what is measured).  Output: per (kernel, grid) average duration for every setting.
export TMPDIR=/tmp; mkdir -p gpurun_out
for S in none pwf:1 pwf:2 pwf:3 dwf:1 dwf:2 dwf:3 pwb:1 pwb:2 dwb:1 dwb:2 dwb:3 dwb:4; do
  rm -rf gpurun_out/ph_prof
  if [ "$S" = none ]; then unset RN_MB_DBG; else export RN_MB_DBG=$S; fi
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ph_prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nms --no-roofline --no-extras > gpurun_out/ph.log 2>&1
  TRACE=$(find gpurun_out/ph_prof -name "bench_kernel_trace.csv" | head -1)
  echo "== $S"; python tools/by_grid.py $TRACE | grep "mb_" | awk '{printf "%s %s %s g%s avg %s\n", $1,$2,$3,$(NF-5),$(NF-3)}' | sort | head -60
done > gpurun_out/r03g_phases.txt 2>&1
rm -rf gpurun_out/ph_prof
