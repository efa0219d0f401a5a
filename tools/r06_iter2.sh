#!/bin/bash
# round 6, second GPU pass: the new parity tests (oracle legs are slow), the chain's backward launches WITHOUT the deferred tower
# weight gradients beside them (what each launch costs alone)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/test_gpu_parity_r6.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r06_i2_tests.log
rm -rf gpurun_out/r06_i2_prof
RN_DEFER_WGRAD=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06_i2_prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-nms --no-roofline > gpurun_out/r06_i2_prof.log 2>&1
TRACE=$(find gpurun_out/r06_i2_prof -name "bench_kernel_trace.csv" | head -1)
python tools/mb_bwd_layers.py $TRACE > gpurun_out/r06_i2_mb_bwd_layers_nodefer.txt
python tools/timeline.py $TRACE > gpurun_out/r06_i2_timeline_nodefer.txt
cp $TRACE gpurun_out/r06_i2_kernel_trace_nodefer.csv
rm -rf gpurun_out/r06_i2_prof
tail -15 gpurun_out/r06_i2_tests.log
