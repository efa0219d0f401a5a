#!/bin/bash
# decode + NMS evidence only (part of round_end.sh): by-grid kernel tables of the hot and the all-candidates input + the bench's nms object
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp; R=${ROUND:-r04}
rm -rf gpurun_out/${R}_nms_prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_nms_prof -o nms -- python tools/nms_prof.py > gpurun_out/${R}_nms.log 2>&1
python tools/by_grid.py $(find gpurun_out/${R}_nms_prof -name "nms_kernel_trace.csv" | head -1) > gpurun_out/${R}_nms_kernels_by_grid.txt
rm -rf gpurun_out/${R}_nms_prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_nms_prof -o nms -- python tools/nms_prof.py stress > gpurun_out/${R}_nms_stress.log 2>&1
python tools/by_grid.py $(find gpurun_out/${R}_nms_prof -name "nms_kernel_trace.csv" | head -1) > gpurun_out/${R}_nms_stress_kernels_by_grid.txt
rm -rf gpurun_out/${R}_nms_prof
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['nms']))" > gpurun_out/${R}_nms_line.json
cat gpurun_out/${R}_nms_line.json; head -9 gpurun_out/${R}_nms_kernels_by_grid.txt | cut -c1-130
