cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x -s -k "winograd_forward_keeps" 2>&1 | tail -15 > gpurun_out/t2.log
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 >> gpurun_out/t2.log
