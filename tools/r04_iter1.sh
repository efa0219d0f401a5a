#!/bin/bash
# round-4 first GPU pass: the new tests, the mAP probe, the extended bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_train_cli.py tests/test_gpu_dist.py "tests/test_gpu_model.py::test_twenty_step_loss_curve_matches_oracle" -q -x -s 2>&1 | tail -40) > gpurun_out/i1_tests_a.log
(timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "loss or dataset or resize or flip" 2>&1 | tail -15) > gpurun_out/i1_tests_b.log
(timeout 900 python -m pytest "tests/test_gpu_fullsize.py::test_cfg5_fp16_detections_match_the_fp32_oracle" -q -x -s 2>&1 | tail -30) > gpurun_out/i1_tests_c.log
timeout 300 python tools/map_probe.py bce_dice > gpurun_out/i1_map_bce.log 2>&1
timeout 300 python tools/map_probe.py focal > gpurun_out/i1_map_focal.log 2>&1
timeout 900 python bench.py > gpurun_out/i1_bench.log 2>&1
tail -5 gpurun_out/i1_tests_a.log; tail -3 gpurun_out/i1_tests_b.log; tail -5 gpurun_out/i1_tests_c.log; tail -3 gpurun_out/i1_map_bce.log; tail -c 1500 gpurun_out/i1_bench.log
