#!/bin/bash
# fp16 head towers: GroupNorm statistics from the multi-level conv's epilogue
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_f16.py -x -q 2>&1 | tail -6
for i in 1 2 3; do for v in "RN_F16_HEAD_EPILOGUE_STATS=0" "RN_F16_HEAD_EPILOGUE_STATS=1"; do
  echo "$v: $(env $v timeout 600 python tools/bench_inference.py 2>&1 | tail -1 | cut -c60-200)"
done; done
