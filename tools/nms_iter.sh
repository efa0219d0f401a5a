export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_kats.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -q -k "detect or nms or cfg5_full or decode or segment_sort" 2>&1 | tail -4) > gpurun_out/r03s.log
python - > gpurun_out/r03s_nms.log 2>&1 <<PY
import sys, json
sys.path[:0]=["/root/repo", "/root/repo/retinanet-tensorflow_amd", "/root/repo/tests"]
import torch, bench
torch.cuda.set_device(0)
r = bench.nms_benchmark(torch.device("cuda:0"))
print(json.dumps({k: r[k] for k in ("hot1pct", "stress")}))
PY
rm -rf gpurun_out/nmsprof; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/nmsprof -o nms -- python tools/nms_prof.py > /dev/null 2>&1
python tools/by_grid.py $(find gpurun_out/nmsprof -name "nms_kernel_trace.csv" | head -1) | grep -E "det_|zero" > gpurun_out/r03s_nms_kernels.txt; rm -rf gpurun_out/nmsprof
cat gpurun_out/r03s.log; tail -1 gpurun_out/r03s_nms.log | cut -c1-500; cat gpurun_out/r03s_nms_kernels.txt | cut -c1-130
