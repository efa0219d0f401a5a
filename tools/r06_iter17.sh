#!/bin/bash
# cfg 5 fp16: the box subnet started a fraction of a layer after the class subnet (the two towers otherwise run in lock step:
# both convs together, then both GroupNorm passes together -- gpurun_out/r06_inf_chron.txt)
export TMPDIR=/tmp; mkdir -p gpurun_out
for i in 1 2; do for v in "RN_NOP=1" "RN_HEADS_OFFSET_US=300" "RN_HEADS_OFFSET_US=450" "RN_HEADS_OFFSET_US=600"; do
  echo "$v: $(env $v timeout 600 python tools/bench_inference.py 2>&1 | tail -1 | cut -c1-230)"
done; done | tee gpurun_out/r06_iter17.log
