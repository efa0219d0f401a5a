#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/ab.sh 2 "RN_HEADS_OFFSET_US=0" "RN_HEADS_OFFSET_US=15" "RN_HEADS_OFFSET_US=30" "RN_HEADS_OFFSET_US=45" "RN_HEADS_OFFSET_US=60" > gpurun_out/i22_ab.log 2>&1
cat gpurun_out/i22_ab.log
