"""Group a rocprofv3 kernel trace by (kernel, blocks): separates the shapes that share a kernel name (host-side).
usage: python tools/by_grid.py <kernel_trace.csv> > profiles/rNN_bench_kernel_trace_by_grid.txt"""
import csv
import os
import re
import sys
from collections import defaultdict


def main(path):
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', ''))[:60]
        blocks = (int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)) * max(int(r['Grid_Size_Y']), 1) // max(int(r['Workgroup_Size_Y']), 1)
        agg[(name, blocks)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from src_hash import kernel_source_hash
    print("# kernel_source_sha: %s" % kernel_source_hash())
    print("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline ; "
          "dispatches grouped by (kernel, blocks)")
    print("%-62s %8s %7s %9s %9s %9s %10s" % ("kernel", "blocks", "calls", "avg_us", "min_us", "max_us", "total_ms"))
    for (name, blocks), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-62s %8d %7d %9.1f %9.1f %9.1f %10.2f" % (name, blocks, len(d), sum(d) / len(d), min(d), max(d), sum(d) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
