#!/bin/bash
# Phase timing of the depthwise chain kernels inside the real step (RN_MB_DBG=dwf:<n> / dwb:<n>; results invalid, durations measured).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp; mkdir -p gpurun_out
for S in none dwf:1 dwf:3 dwb:1 dwb:4; do
  rm -rf gpurun_out/ph_prof
  if [ "$S" = none ]; then unset RN_MB_DBG; else export RN_MB_DBG=$S; fi
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ph_prof -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nms --no-roofline --no-extras > gpurun_out/ph.log 2>&1
  TRACE=$(find gpurun_out/ph_prof -name "bench_kernel_trace.csv" | head -1)
  echo "== $S"; python tools/chron.py $TRACE | grep "mb_dw" | awk '{print $2, $5}' | sort -k2 | awk '{a[$2]=a[$2]" "$1} END{for(k in a) print k, a[k]}'
done > gpurun_out/i16_dw_phases.txt 2>&1
rm -rf gpurun_out/ph_prof
cat gpurun_out/i16_dw_phases.txt
