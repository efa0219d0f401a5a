#!/bin/bash
# per-kernel durations of the decode + NMS scan under rocprofv3, cfg-5 hot input: bash tools/scan_ab.sh "ENV_A" "ENV_B" ...
export TMPDIR=/tmp; mkdir -p gpurun_out
for v in "$@"; do
  rm -rf gpurun_out/scanprof
  env $v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/scanprof -o nms -- python tools/nms_prof.py > /dev/null 2>&1
  echo "== $v"; python tools/by_grid.py $(find gpurun_out/scanprof -name "nms_kernel_trace.csv" | head -1) | grep "det_scan" | cut -c1-140
done
rm -rf gpurun_out/scanprof
