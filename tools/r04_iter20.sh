#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_mbchain.py tests/test_gpu_wino_tower.py -q -x --deselect tests/test_gpu_model.py::test_twenty_step_loss_curve_matches_oracle 2>&1 | tail -3) > gpurun_out/i20_tests.log
cat gpurun_out/i20_tests.log
bash tools/ab.sh 3 "RN_X=0" > gpurun_out/i20_ab.log 2>&1
cat gpurun_out/i20_ab.log
TAG=p20 bash tools/r04_prof.sh > /dev/null 2>&1
grep "reduce_rows" gpurun_out/p20_chron.txt
