"""Per-kernel averages of every counter found under the given rocprofv3 --pmc output directories (+ the kernel trace's durations)."""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]]["duration_ns(%s)" % os.path.basename(d)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, cs in acc.items():
    if "conv" not in k and "gemm" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-40s n=%3d avg %.4g" % (c, len(v), sum(v) / len(v)))
