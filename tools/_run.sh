cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench5.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final_prof.log 2>&1
