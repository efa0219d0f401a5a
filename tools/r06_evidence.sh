#!/bin/bash
# round-6 evidence on the final kernel sources: the QUICK round-end set + the product timings and their leave-one-out
QUICK=1 ROUND=r06 bash tools/round_end.sh
cd ${GRAFT_REPO_ROOT:-.}
timeout 300 python tools/x3_bench.py > gpurun_out/r06_x3_products.txt 2>&1
for v in 0 1 2 4 8 15; do echo "RN_X3_DBG=$v $(env RN_X3_DBG=$v timeout 300 python tools/x3_bench.py 2>&1 | grep 'mode 1' | tail -1)"; done > gpurun_out/r06_x3_leave_one_out.txt 2>&1
for v in 0 1 2 4 8 15; do echo "RN_X3_DBG=$v RN_X3_BFRAG=0 $(env RN_X3_DBG=$v RN_X3_BFRAG=0 timeout 300 python tools/x3_bench.py 2>&1 | grep 'mode 1' | tail -1)"; done >> gpurun_out/r06_x3_leave_one_out.txt 2>&1
timeout 300 python tools/bench_inference.py > gpurun_out/r06_inference_lines.txt 2>&1
