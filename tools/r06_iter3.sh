#!/bin/bash
# round 6, third GPU pass: the software-pipelined product kernel (RN_X3_PIPE) -- parity, stand-alone time, the step
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
RN_X3_PIPE=1 timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_wino_tower.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06_i3_tests.log
for p in 0 1 0 1; do echo "RN_X3_PIPE=$p $(RN_X3_PIPE=$p timeout 300 python tools/x3_bench.py 2>&1 | grep 'mode 1' | tail -1)"; done > gpurun_out/r06_i3_x3.txt 2>&1
for p in 0 1 0 1; do echo "RN_X3_PIPE=$p $(RN_X3_PIPE=$p timeout 300 python bench.py --no-cpu-baseline --no-extras --no-nms --no-roofline 2>/dev/null | grep '^{' | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print(r["value"], r["ms_per_step"])')"; done > gpurun_out/r06_i3_step.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity_r6.py -x -q -m gpu -s -k conditioned 2>&1 | grep -v Warning | tail -30 > gpurun_out/r06_i3_cond.log
cat gpurun_out/r06_i3_tests.log gpurun_out/r06_i3_x3.txt gpurun_out/r06_i3_step.txt; tail -5 gpurun_out/r06_i3_cond.log
