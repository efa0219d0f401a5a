#!/bin/bash
# cfg 5 fp16: chronological kernel trace of one batch (heads: which launches overlap, how long each conv takes)
export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/infprof
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/infprof -o inf -- python3 $GRAFT_REPO_ROOT/tools/inf_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r06_inf_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/infprof -name "inf_kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/r06_inf_chron.txt <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# last pass: starts at the last pad_cast_rgb launch
marks = [i for i, r in enumerate(rows) if 'pad_cast_rgb' in r['Kernel_Name']]
lo = marks[-1]
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'^void\s+', '', r['Kernel_Name']).replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)[:70]
    blocks = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])) * max(1, int(r['Grid_Size_Y'])) * max(1, int(r['Grid_Size_Z']))
    print("%9.1f %8.1f q%-3s b%-7d %s" % ((s - t0) / 1e3, (e - s) / 1e3, r['Queue_Id'], blocks, n))
PY
rm -rf gpurun_out/infprof
wc -l gpurun_out/r06_inf_chron.txt
