#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_ops.py tests/test_gpu_kats.py -q -k "nms or detect or decode or cfg5_full" 2>&1 | tail -6) > gpurun_out/i14_tests.log
cat gpurun_out/i14_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/i14_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/i14_bench.log') if x.startswith('{')][-1]
d=json.loads(l); print(d['value']); print(json.dumps(d['nms']['hot1pct'])); print(json.dumps(d['nms']['stress']))
PY
rm -rf gpurun_out/i14_nms_prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/i14_nms_prof -o nms -- python tools/nms_prof.py > gpurun_out/i14_nms.log 2>&1
python tools/by_grid.py $(find gpurun_out/i14_nms_prof -name "nms_kernel_trace.csv" | head -1) > gpurun_out/i14_nms_kernels.txt
rm -rf gpurun_out/i14_nms_prof
grep det_ gpurun_out/i14_nms_kernels.txt | cut -c1-140
