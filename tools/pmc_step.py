"""Per-kernel HBM traffic and matrix-core occupancy of the training step from rocprofv3 PMC passes over bench.py (host-side).
usage: python tools/pmc_step.py <trace_dir> <fetch_dir> <write_dir> <sq_dir> > profiles/rNN_pmc_step.txt
  trace_dir : rocprofv3 --kernel-trace (durations, undisturbed)
  fetch_dir : --pmc FETCH_SIZE            write_dir : --pmc WRITE_SIZE        (separate passes: they do not fit one)
  sq_dir    : --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
Bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads -- an upper bound for kernels whose reads are narrower).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES):
the share of the busy compute units' SIMD cycles with a matrix instruction in flight.  Kernels are grouped by (name, blocks)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

HBM_PEAK = 8000.0   # GB/s
FP32_MFMA_PEAK = 157.3


def short(n):
    n = re.sub(r'^void\s+', '', n).replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*', '', n)[:64]


def key_of(r):
    blocks = (int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)) * max(int(r.get('Grid_Size_Y', 1) or 1), 1) // max(int(r.get('Workgroup_Size_Y', 1) or 1), 1)
    return short(r['Kernel_Name']), blocks


def counters(directory):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            blocks = int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1) if 'Grid_Size' in r else 0
            acc[(short(r['Kernel_Name']), blocks)][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


def main(trace_dir, fetch_dir, write_dir, sq_dir):
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[key_of(r)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    fe, wr, sq = counters(fetch_dir), counters(write_dir), counters(sq_dir)
    rows = []
    for k, d in dur.items():
        if len(d) < 3:
            continue
        us = sorted(d)[len(d) // 2]
        f = fe.get(k, {}).get('FETCH_SIZE')
        w = wr.get(k, {}).get('WRITE_SIZE')
        mb = (2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e6 if f and w else None
        s = sq.get(k, {})
        mf = s.get('SQ_VALU_MFMA_BUSY_CYCLES')
        bc = s.get('SQ_BUSY_CU_CYCLES')
        busy = (sum(mf) / len(mf)) / (4.0 * sum(bc) / len(bc)) if mf and bc and sum(bc) else None
        rows.append((us * len(d), k, len(d), us, mb, busy))
    rows.sort(reverse=True)
    print("%-66s %7s %6s %9s %9s %9s %7s %9s" % ("kernel", "blocks", "calls", "median_us", "HBM_MB", "GB/s", "of_8TB", "MFMA_busy"))
    for tot, (name, blocks), calls, us, mb, busy in rows[:70]:
        gbs = mb / us * 1e3 if mb else None
        print("%-66s %7d %6d %9.1f %9s %9s %7s %9s" % (name, blocks, calls, us, "%.2f" % mb if mb is not None else "-",
                                                        "%.0f" % gbs if gbs else "-", "%.2f" % (gbs / HBM_PEAK) if gbs else "-",
                                                        "%.2f" % busy if busy is not None else "-"))


if __name__ == "__main__":
    main(*sys.argv[1:5])
