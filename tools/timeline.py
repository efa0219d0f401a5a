"""Where a step's time goes, from a rocprofv3 kernel trace of `bench.py` (host-side analysis, no GPU needed):
usage: python tools/timeline.py gpurun_out/final_prof/bench_kernel_trace.csv
Takes the step of median wall time among the last ten (from one anchor-assignment kernel to the next), and reports: wall time, time with >= 1 kernel
running, idle time, time with two kernels overlapping, kernels per step, and the per-kernel-family totals."""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'^void\s+', '', n).replace('(anonymous namespace)::', '').replace('at::native::', '')
    return re.sub(r'[<(].*', '', n)


def main(path):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'assign_kernel' in r['Kernel_Name']]
    # steps = spans between consecutive assign kernels; of the last ten, the one of median WALL time (the profiler's buffer
    # flushes put multi-millisecond holes into some steps)
    spans = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)]
    spans = [s for s in spans if 100 < s[1] - s[0] < 700][-10:]
    spans.sort(key=lambda s: int(rows[s[1]]['Start_Timestamp']) - int(rows[s[0]]['Start_Timestamp']))
    lo, hi = spans[len(spans) // 2]
    step = rows[lo:hi]
    t0 = int(step[0]['Start_Timestamp'])
    t1 = int(rows[hi]['Start_Timestamp'])
    ev = []
    for r in step:
        ev.append((int(r['Start_Timestamp']), 1))
        ev.append((int(r['End_Timestamp']), -1))
    ev.sort()
    depth, last, busy, over = 0, t0, 0, 0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    fam = defaultdict(lambda: [0, 0.0])
    for r in step:
        k = short(r['Kernel_Name'])
        fam[k][0] += 1
        fam[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    wall = (t1 - t0) / 1e3
    print("step: %d kernels, wall %.0f us, >=1 kernel running %.0f us (%.0f%%), idle %.0f us (%.0f%%), two overlapping %.0f us"
          % (len(step), wall, busy / 1e3, 100 * busy / 1e3 / wall, wall - busy / 1e3, 100 * (wall - busy / 1e3) / wall, over / 1e3))
    print("queues:", sorted(set(r['Queue_Id'] for r in step)))
    # where the GPU is idle: the longest stretches with no kernel running, with the kernels on either side
    gaps, cur_end, prev = [], t0, None
    for r in sorted(step + [rows[hi]], key=lambda r: int(r['Start_Timestamp'])):
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if st > cur_end and prev is not None:
            gaps.append((st - cur_end, (cur_end - t0) / 1e3, short(prev['Kernel_Name']), short(r['Kernel_Name'])))
        if en > cur_end:
            cur_end, prev = en, r
    gaps.sort(reverse=True)
    print("idle stretches: %d, the longest:" % len(gaps))
    for g, at, a_, b_ in gaps[:8]:
        print("  %6.1f us at %7.1f us: after %s, before %s" % (g / 1e3, at, a_, b_))
    tot = sum(v[1] for v in fam.values())
    print("sum of kernel durations %.0f us" % tot)
    for k, (n, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("  %-34s %4d launches %8.1f us  %5.1f%%" % (k, n, us, 100 * us / tot))


if __name__ == "__main__":
    main(sys.argv[1])
