#!/bin/bash
# env-switch sweep of the headline step on one box (each arm: bench.py --steps 150, no extras)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
run() { echo "$1: $(env $1 timeout 300 python bench.py --steps 150 --no-cpu-baseline --no-extras --no-nms --no-roofline 2>/dev/null | grep '^{' | python -c 'import json,sys; r=json.loads(sys.stdin.readline()); print(r["value"], r["ms_per_step"])')"; }
{
run "RN_NONE=0"
run "RN_HEADS_TWO_STREAMS=0"
run "RN_FPN_TWO_STREAMS=0"
run "RN_SIDE_PRIO=0"
run "RN_X3_NST=1"
run "RN_X3_TILE=64"
run "RN_NONE=0"
run "RN_DEFER_SIDE=0"
run "RN_MB_PW_BIG_BLOCKS=1024"
run "RN_MB_DWB_TH=16"
run "RN_OCC=0"
run "RN_MB_WGRAD_FIRST=0"
run "RN_NONE=0"
} > gpurun_out/r06_sweep.txt 2>&1
cat gpurun_out/r06_sweep.txt
