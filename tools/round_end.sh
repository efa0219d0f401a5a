#!/bin/bash
# Round-end evidence run (GPU box): full GPU test suite, smoke, the bench line, its rocprofv3 summary, PMC traffic of the
# roofline kernels, the other configs (QUICK=1: bench, profile and PMC passes only).  Summaries are copied into profiles/ by hand afterwards (see profiles/README.md).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${ROUND:-r06}
[ -z "$QUICK" ] && timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/${R}_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/${R}_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/${R}_bench.log 2>&1
grep '^{"metric"' gpurun_out/${R}_bench.log | tail -1 > gpurun_out/${R}_bench_line.json
timeout 600 python bench.py --gpus 1 --spawn --force-collective --no-cpu-baseline --no-nms --no-roofline --no-extras 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/${R}_bench_line_forced_collective.json
rm -rf gpurun_out/${R}_prof gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_prof -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/${R}_prof.log 2>&1
TRACE=$(find gpurun_out/${R}_prof -name "bench_kernel_trace.csv" | head -1)
python tools/timeline.py $TRACE > gpurun_out/${R}_bench_step_timeline.txt
python tools/mb_bwd_layers.py $TRACE > gpurun_out/${R}_mb_bwd_layers.txt
python tools/by_grid.py $TRACE > gpurun_out/${R}_bench_kernel_trace_by_grid.txt
cp $(find gpurun_out/${R}_prof -name "bench_kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_pmc_fetch -- python tools/gemm_pmc.py > gpurun_out/${R}_pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_pmc_write -- python tools/gemm_pmc.py > gpurun_out/${R}_pmc_w.log 2>&1
python tools/pmc_traffic.py gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write > gpurun_out/${R}_pmc_traffic.json
ROUND=$R bash tools/pmc_step.sh > gpurun_out/${R}_pmc_step.log 2>&1
[ -z "$QUICK" ] && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_inf_prof -o inf -- python tools/inf_prof.py > gpurun_out/${R}_inf_prof.log 2>&1
[ -z "$QUICK" ] && cp $(find gpurun_out/${R}_inf_prof -name "inf_kernel_stats.csv" | head -1) gpurun_out/${R}_inference_f16_kernel_stats.csv
rm -rf gpurun_out/${R}_inf_prof
[ -z "$QUICK" ] && timeout 200 python tools/tower_bench.py 720 > gpurun_out/${R}_tower.log 2>&1
rm -rf gpurun_out/${R}_nms_prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_nms_prof -o nms -- python tools/nms_prof.py > gpurun_out/${R}_nms.log 2>&1
python tools/by_grid.py $(find gpurun_out/${R}_nms_prof -name "nms_kernel_trace.csv" | head -1) > gpurun_out/${R}_nms_kernels_by_grid.txt
rm -rf gpurun_out/${R}_nms_prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}_nms_prof -o nms -- python tools/nms_prof.py stress > gpurun_out/${R}_nms_stress.log 2>&1
python tools/by_grid.py $(find gpurun_out/${R}_nms_prof -name "nms_kernel_trace.csv" | head -1) > gpurun_out/${R}_nms_stress_kernels_by_grid.txt
rm -rf gpurun_out/${R}_nms_prof gpurun_out/${R}_prof gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write
[ -z "$QUICK" ] && (hipcc --offload-arch=gfx950 -O3 tools/micro/boundary.hip -o /tmp/boundary && /tmp/boundary > gpurun_out/${R}_micro_boundary.txt 2>&1)
[ -z "$QUICK" ] && (hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_cluster.hip -o /tmp/xcd_cluster && timeout 300 /tmp/xcd_cluster > gpurun_out/${R}_micro_xcd_cluster.txt 2>&1)
[ -z "$QUICK" ] && (RN_MB_RESIDENT=1 timeout 300 python tools/mb_resident_phases.py 512 2 > gpurun_out/${R}_mb_resident_phases.txt 2>&1)
[ -z "$QUICK" ] && (timeout 300 python tools/x3_bench.py > gpurun_out/${R}_x3_products.txt 2>&1; for v in 0 1 2 4 8 15; do echo "RN_X3_DBG=$v $(env RN_X3_DBG=$v timeout 300 python tools/x3_bench.py 2>&1 | grep 'mode 1' | tail -1)"; done > gpurun_out/${R}_x3_leave_one_out.txt 2>&1)
[ -z "$QUICK" ] && (timeout 600 python tools/f16_trained_probe.py 2500 1e-2 2>&1 | tail -8 > gpurun_out/${R}_f16_trained_probe.txt)
[ -z "$QUICK" ] && (timeout 400 python tools/f16_layer_probe.py > gpurun_out/${R}_f16_layers.txt 2>&1; timeout 300 python tools/bench_inference.py > gpurun_out/${R}_inference_lines.txt 2>&1)
tail -3 gpurun_out/${R}_tests.log; tail -1 gpurun_out/${R}_smoke.log; cut -c1-400 gpurun_out/${R}_bench_line.json
