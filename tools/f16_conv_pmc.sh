#!/bin/bash
# counters of the fp16 head conv alone (GPU box): MFMA busy, LDS conflicts / waits, effective clock
export TMPDIR=/tmp; mkdir -p gpurun_out; T=${TAG:-f16pmc}
rm -rf gpurun_out/${T}_*
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${T}_a -- python tools/f16_conv_one.py > gpurun_out/${T}_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/${T}_b -- python tools/f16_conv_one.py > gpurun_out/${T}_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_DATA_FIFO_FULL --output-format csv -d gpurun_out/${T}_c -- python tools/f16_conv_one.py > gpurun_out/${T}_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/${T}_d -- python tools/f16_conv_one.py > gpurun_out/${T}_d.log 2>&1
python tools/pmc_dump.py gpurun_out/${T}_a gpurun_out/${T}_b gpurun_out/${T}_c gpurun_out/${T}_d | tee gpurun_out/${T}.txt
rm -rf gpurun_out/${T}_a gpurun_out/${T}_b gpurun_out/${T}_c gpurun_out/${T}_d
python tools/f16_head_bench.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/${T}.txt
python tools/gemm_bench.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/${T}.txt
