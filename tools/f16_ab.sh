#!/bin/bash
# fp16 head conv alone + cfg-5 inference, A/B of env settings: bash tools/f16_ab.sh "ENV_A" "ENV_B" ...
export TMPDIR=/tmp; mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"; env $v python tools/f16_head_bench.py 2>&1 | grep -v amdgpu
done | tee gpurun_out/f16_ab.log
for i in 1 2; do for v in "$@"; do
  echo "$v: $(env $v timeout 600 python tools/bench_inference.py 2>&1 | tail -1 | cut -c1-200)"
done; done | tee -a gpurun_out/f16_ab.log
