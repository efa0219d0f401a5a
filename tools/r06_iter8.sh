#!/bin/bash
# round 6: conv -> dropout -> statistics as one launch (DenseNet) -- parity, cfg 4 with and without; then the whole GPU suite
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_dropout.py tests/test_gpu_backbones.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_i8_tests.log
for v in 0 1 0 1; do
  echo "RN_DENSE_FUSED_DROPOUT=$v $(RN_DENSE_FUSED_DROPOUT=$v timeout 600 python tools/bench_configs.py densenet_121 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print("cfg4", r["images_per_sec"], r["ms_per_step"], r["peak_mem_GB"])')"
done > gpurun_out/r06_i8_cfgs.txt 2>&1
cat gpurun_out/r06_i8_tests.log gpurun_out/r06_i8_cfgs.txt
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06_fulltests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06_smoke.log 2>&1
tail -6 gpurun_out/r06_fulltests.log; tail -1 gpurun_out/r06_smoke.log
