# cfg 5 inference: fp16 with / without the folded GroupNorms, then a kernel-stat profile of the fp16 pass
export TMPDIR=/tmp; mkdir -p gpurun_out
RN_F16_FOLD=0 timeout 400 python tools/bench_inference.py 2>&1 | tail -1 > gpurun_out/r03_inf_nofold.log
RN_F16_FOLD=1 timeout 400 python tools/bench_inference.py 2>&1 | tail -2 > gpurun_out/r03_inf_fold.log
rm -rf gpurun_out/infprof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/infprof -o inf -- python tools/inf_prof.py > gpurun_out/r03_inf_prof.log 2>&1
cp $(find gpurun_out/infprof -name "inf_kernel_stats.csv" | head -1) gpurun_out/r03_inference_f16_kernel_stats.csv; rm -rf gpurun_out/infprof
cat gpurun_out/r03_inf_nofold.log gpurun_out/r03_inf_fold.log; head -14 gpurun_out/r03_inference_f16_kernel_stats.csv | cut -c1-150
