#!/bin/bash
# PMC evidence of the training step (GPU box): kernel trace + three counter passes over the same bench command.
export TMPDIR=/tmp; mkdir -p gpurun_out
R=${ROUND:-r06}
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-nms --no-roofline --no-extras --no-graph"
rocprofv3 -L > gpurun_out/${R}_counters_list.txt 2>&1
for P in trace fetch write sq; do rm -rf gpurun_out/${R}_pmc_$P; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_trace -- $CMD > gpurun_out/${R}_pmc_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_pmc_fetch -- $CMD > gpurun_out/${R}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_pmc_write -- $CMD > gpurun_out/${R}_pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${R}_pmc_sq -- $CMD > gpurun_out/${R}_pmc_sq.log 2>&1
python tools/pmc_step.py gpurun_out/${R}_pmc_trace gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write gpurun_out/${R}_pmc_sq > gpurun_out/${R}_pmc_step.txt 2> gpurun_out/${R}_pmc_step.err
head -3 $(find gpurun_out/${R}_pmc_sq -name "*counter_collection.csv" | head -1) > gpurun_out/${R}_pmc_sq_head.txt 2>&1
grep -i -E "MFMA|BUSY" gpurun_out/${R}_counters_list.txt | head -40 > gpurun_out/${R}_counters_mfma.txt
for P in trace fetch write sq; do rm -rf gpurun_out/${R}_pmc_$P; done
head -30 gpurun_out/${R}_pmc_step.txt; cat gpurun_out/${R}_pmc_step.err | tail -5; cat gpurun_out/${R}_pmc_sq_head.txt
