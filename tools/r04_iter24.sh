#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
P=$PWD/retinanet-tensorflow_amd
bash tools/ab.sh 3 "RN_LIB_PATH=$P/librn_hip.so" "RN_LIB_PATH=$P/librn_hip_occ.so" > gpurun_out/i24_ab.log 2>&1
cat gpurun_out/i24_ab.log
RN_LIB_PATH=$P/librn_hip_occ.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('occ', d['value'], r['frac'], r['kernel_ms'], [ (e['frac'], e['kernel_ms']) for e in r['entries']])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-nms --no-extras 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('base', d['value'], r['frac'], r['kernel_ms'], [ (e['frac'], e['kernel_ms']) for e in r['entries']])"
