"""Per-kernel-family totals of a rocprofv3 kernel trace (host-side): usage: python tools/family.py <kernel_trace.csv> [skip_first_fraction]"""
import csv, re, sys
from collections import defaultdict


def short(n):
    n = re.sub(r'^void\s+', '', n).replace('(anonymous namespace)::', '').replace('at::native::', '')
    return re.sub(r'[<(].*', '', n)


rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]                 # the later part of the run: steady-state steps
fam = defaultdict(lambda: [0, 0.0])
for r in rows:
    k = short(r['Kernel_Name'])
    fam[k][0] += 1
    fam[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in fam.values())
print("# %d dispatches, %.1f ms of kernel time (last %.0f %% of the trace)" % (len(rows), tot / 1e3, 100 * (1 - skip)))
for k, (c, us) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %-40s %6d launches %10.1f us  %5.1f%%  %7.1f us each" % (k, c, us, 100 * us / tot, us / c))
