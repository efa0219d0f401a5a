#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/ab.sh 2 "RN_PROD_LDS_PAD_BWD=0" "RN_PROD_LDS_PAD_BWD=16384" "RN_PROD_LDS_PAD_BWD=30000" "RN_PROD_LDS_PAD_FWD=16384" "RN_PROD_LDS_PAD_FWD=34000" "RN_PROD_LDS_PAD_FWD=16384 RN_PROD_LDS_PAD_BWD=16384" > gpurun_out/i12_ab.log 2>&1
cat gpurun_out/i12_ab.log
