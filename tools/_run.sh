cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4b -o c4 -- python tools/prof_cfg.py densenet_121 640 4 > gpurun_out/prof_c4b.log 2>&1
