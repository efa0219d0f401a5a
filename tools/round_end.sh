#!/bin/bash
# Round-end evidence run (GPU box): full GPU test suite, smoke, the bench line, its rocprofv3 summary, the other configs.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/final_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/final_smoke.log 2>&1
timeout 400 python bench.py > gpurun_out/final_bench.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final_prof.log 2>&1
timeout 600 python tools/bench_configs.py > gpurun_out/final_cfgs.log 2>&1
timeout 400 python tools/bench_inference.py > gpurun_out/final_inf.log 2>&1
