#!/bin/bash
# round 6: DenseNet -- cached per-channel statistic rows; parity, cfg 4 A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_backbones.py tests/test_gpu_dropout.py tests/test_gpu_x3.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06_i11_tests.log
timeout 1500 python -m pytest tests/test_gpu_parity_r6.py tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg4 or densenet" 2>&1 | tail -6 >> gpurun_out/r06_i11_tests.log
for v in 0 1 0 1; do
  echo "RN_DENSE_CACHED_ROWS=$v $(RN_DENSE_CACHED_ROWS=$v timeout 600 python tools/bench_configs.py densenet_121 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print("cfg4", r["images_per_sec"], r["ms_per_step"])')"
done > gpurun_out/r06_i11_cfgs.txt 2>&1
cat gpurun_out/r06_i11_tests.log gpurun_out/r06_i11_cfgs.txt
