#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_mbchain.py -q -x -s 2>&1 | tail -30) > gpurun_out/i5_tests.log
cat gpurun_out/i5_tests.log | tail -8
bash tools/ab.sh 3 "RN_MB_PW_BIG=0" "RN_MB_PW_BIG=1" "RN_MB_PW_BIG=1 RN_MB_PW_BIG_BLOCKS=1024" "RN_MB_PW_BIG=1 RN_MB_PW_BIG_BLOCKS=256" > gpurun_out/i5_ab.log 2>&1
cat gpurun_out/i5_ab.log
rm -rf gpurun_out/i5_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/i5_prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-roofline --no-extras > gpurun_out/i5_prof.log 2>&1
TRACE=$(find gpurun_out/i5_prof -name "bench_kernel_trace.csv" | head -1)
python tools/timeline.py $TRACE > gpurun_out/i5_timeline.txt; python tools/chron.py $TRACE > gpurun_out/i5_chron.txt
rm -rf gpurun_out/i5_prof
grep -n "mb_pw_bwd\|mb_dw_bwd" gpurun_out/i5_chron.txt | tail -14
