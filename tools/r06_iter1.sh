#!/bin/bash
# round 6, first GPU pass: the new multi-rank step on one rank, the TF known-answer vectors, a bench line, a kernel trace
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_kats.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r06_i1_tests.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -15 >> gpurun_out/r06_i1_tests.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r06_i1_bench.log 2>&1
timeout 600 python bench.py --gpus 1 --spawn --force-collective --no-cpu-baseline --no-nms --no-roofline --no-extras > gpurun_out/r06_i1_bench_fc.log 2>&1
rm -rf gpurun_out/r06_i1_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_i1_prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-nms --no-roofline > gpurun_out/r06_i1_prof.log 2>&1
TRACE=$(find gpurun_out/r06_i1_prof -name "bench_kernel_trace.csv" | head -1)
python tools/timeline.py $TRACE > gpurun_out/r06_i1_timeline.txt
python tools/by_grid.py $TRACE > gpurun_out/r06_i1_by_grid.txt
cp $TRACE gpurun_out/r06_i1_kernel_trace.csv
rm -rf gpurun_out/r06_i1_prof
tail -5 gpurun_out/r06_i1_tests.log; tail -1 gpurun_out/r06_i1_bench.log | cut -c1-600; tail -1 gpurun_out/r06_i1_bench_fc.log | cut -c1-300
