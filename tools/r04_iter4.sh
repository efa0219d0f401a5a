#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/f16_error_probe.py 384 > gpurun_out/i4_f16_probe.log 2>&1
bash tools/ab.sh 3 "RN_WINO_GN_W=2" "RN_WINO_GN_W=1" "RN_WINO_GN_W=4" > gpurun_out/i4_ab.log 2>&1
grep -v amdgpu gpurun_out/i4_f16_probe.log | tail -4; cat gpurun_out/i4_ab.log
