#!/bin/bash
# Phase timing of mb_pw_bwd_big_kernel inside the real step (RN_MB_DBG=pwB:<phase>: the kernel returns after that phase; results invalid).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp; mkdir -p gpurun_out
for S in none pwB:1 pwB:2 pwB:3; do
  rm -rf gpurun_out/ph_prof
  if [ "$S" = none ]; then unset RN_MB_DBG; else export RN_MB_DBG=$S; fi
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ph_prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nms --no-roofline --no-extras > gpurun_out/ph.log 2>&1
  TRACE=$(find gpurun_out/ph_prof -name "bench_kernel_trace.csv" | head -1)
  echo "== $S"; python tools/by_grid.py $TRACE | grep "big" | cut -c1-160
done > gpurun_out/i6_phases.txt 2>&1
rm -rf gpurun_out/ph_prof
cat gpurun_out/i6_phases.txt
