#!/bin/bash
# round-4 GPU pass 2: fixed tests, fp16 error probe, focal at lr 1e-3, chain A/B (wgrad-first, wgrad splits, stride-templated dw_bwd), full bench
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_train_cli.py tests/test_gpu_mbchain.py -q -x -s 2>&1 | tail -40) > gpurun_out/i2_tests_a.log
(timeout 900 python -m pytest "tests/test_gpu_fullsize.py::test_cfg5_fp16_detections_match_the_fp32_oracle" -q -x -s 2>&1 | tail -60) > gpurun_out/i2_tests_c.log
timeout 600 python tools/f16_error_probe.py 384 > gpurun_out/i2_f16_probe.log 2>&1
RN_F16_FOLD=0 timeout 600 python tools/f16_error_probe.py 384 > gpurun_out/i2_f16_probe_nofold.log 2>&1
timeout 300 python tools/map_probe.py focal 1e-3 > gpurun_out/i2_map_focal_lr3.log 2>&1
bash tools/ab.sh 3 "RN_MB_WGRAD_FIRST=0" "RN_MB_WGRAD_FIRST=1" "RN_MB_WGRAD_FIRST=1 RN_MB_WGRAD_BIG_BLOCKS=1024" "RN_MB_WGRAD_FIRST=1 RN_MB_WGRAD_BIG_BLOCKS=2048" > gpurun_out/i2_ab.log 2>&1
rm -rf gpurun_out/i2_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/i2_prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-roofline --no-extras > gpurun_out/i2_prof.log 2>&1
TRACE=$(find gpurun_out/i2_prof -name "bench_kernel_trace.csv" | head -1)
python tools/timeline.py $TRACE > gpurun_out/i2_timeline.txt; python tools/chron.py $TRACE > gpurun_out/i2_chron.txt
rm -rf gpurun_out/i2_prof
timeout 900 python bench.py > gpurun_out/i2_bench.log 2>&1
tail -5 gpurun_out/i2_tests_a.log; tail -5 gpurun_out/i2_tests_c.log; cat gpurun_out/i2_ab.log; tail -c 3000 gpurun_out/i2_bench.log
