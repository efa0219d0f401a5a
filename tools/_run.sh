cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "depthwise or tiny_and_ragged" 2>&1 | tail -3 > gpurun_out/t5.log
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -3 >> gpurun_out/t5.log
timeout 300 python bench.py --no-cpu-baseline --no-nms > gpurun_out/bench6.log 2>&1
