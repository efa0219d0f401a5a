#!/bin/bash
# round 6: the fallback multi-rank path keeps the deferred tower weight gradients -- dist tests, rate of both paths on one rank
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06_i12_tests.log
for i in 1 2; do
timeout 600 python bench.py --gpus 1 --spawn --force-collective --no-cpu-baseline --no-nms --no-roofline --no-extras 2>/dev/null | grep '^{"metric"' | tail -1 | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); a=r["config"]["allreduce"]; print("captured", r["value"], "fallback", a["one_graph_per_part_eager_collectives"])'
done > gpurun_out/r06_i12_fc.txt 2>&1
cat gpurun_out/r06_i12_tests.log gpurun_out/r06_i12_fc.txt
