export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_mbchain.py -q -x 2>&1 | tail -15) > gpurun_out/r03f_mb.log
rm -rf gpurun_out/r03f_prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03f_prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-roofline --no-extras > gpurun_out/r03f_prof.log 2>&1
TRACE=$(find gpurun_out/r03f_prof -name "bench_kernel_trace.csv" | head -1)
python tools/timeline.py $TRACE > gpurun_out/r03f_timeline.txt; python tools/chron.py $TRACE > gpurun_out/r03f_chron.txt
rm -rf gpurun_out/r03f_prof
timeout 300 python bench.py --no-cpu-baseline --no-nms --no-roofline --no-extras > gpurun_out/r03f_bench.log 2>&1
tail -3 gpurun_out/r03f_mb.log; head -12 gpurun_out/r03f_timeline.txt; tail -1 gpurun_out/r03f_bench.log | cut -c1-160
