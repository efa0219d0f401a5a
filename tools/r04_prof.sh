#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${TAG:-px}
rm -rf gpurun_out/${T}_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-roofline --no-extras > gpurun_out/${T}_prof.log 2>&1
TRACE=$(find gpurun_out/${T}_prof -name "bench_kernel_trace.csv" | head -1)
python tools/timeline.py $TRACE > gpurun_out/${T}_timeline.txt; python tools/chron.py $TRACE > gpurun_out/${T}_chron.txt; python tools/by_grid.py $TRACE > gpurun_out/${T}_bygrid.txt
rm -rf gpurun_out/${T}_prof
head -12 gpurun_out/${T}_timeline.txt
