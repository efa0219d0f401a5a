#!/bin/bash
# cfg 5 fp16: A/B of the head statistics / 3x3-fold switches at today's kernel speeds
export TMPDIR=/tmp; mkdir -p gpurun_out
for i in 1 2; do for v in "RN_NOP=1" "RN_F16_HEAD_STATS=1" "RN_F16_HEAD_STATS=1 RN_F16_HEAD_STATS_MIN_ROWS=4096" "RN_F16_FOLD_3X3=1"; do
  echo "$v: $(env $v timeout 600 python tools/bench_inference.py 2>&1 | tail -1 | cut -c1-230)"
done; done | tee gpurun_out/r06_iter15.log
