#!/bin/bash
# after moving the fragment-operand kernel into its own translation unit: tests, the cfg 3 / cfg 4 steps against the tree of 0104ad0 (_old), headline A/B
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_wino_tower.py -x -q 2>&1 | tail -3
for i in 1 2; do for d in _old .; do for c in resnet_50 densenet_121; do echo "$d $c: $(cd $R/$d && timeout 600 python tools/bench_configs.py $c 2>/dev/null | cut -c1-120)"; done; done; done
for i in 1 2; do for v in "RN_X3_BFRAG=0" "RN_X3_BFRAG=1"; do
  echo "$v: $(env $v timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120)"
done; done
python tools/x3_bench.py 2>&1 | tail -3
