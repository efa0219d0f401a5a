#!/bin/bash
# the two PMC passes over the roofline kernels alone -> gpurun_out/${ROUND}_pmc_traffic.json (part of round_end.sh)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; export TMPDIR=/tmp; R=${ROUND:-r04}
rm -rf gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_pmc_fetch -- python tools/gemm_pmc.py > gpurun_out/${R}_pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_pmc_write -- python tools/gemm_pmc.py > gpurun_out/${R}_pmc_w.log 2>&1
python tools/pmc_traffic.py gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write > gpurun_out/${R}_pmc_traffic.json
head -3 $(find gpurun_out/${R}_pmc_fetch -name "*counter_collection.csv" | head -1) | cut -c1-300
rm -rf gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write
cat gpurun_out/${R}_pmc_traffic.json
