"""Per-launch table of the MobileNetV2 chain's BACKWARD pass ("start from evidence") from a rocprofv3
kernel trace of bench.py (host-side analysis, no GPU needed).
usage: python tools/mb_bwd_layers.py <bench_kernel_trace.csv> > profiles/r06_mb_bwd_layers.txt

The chain's backward launches of one step come in a fixed order (ops_mb._MbChain.backward): the tail 1x1 conv, then for every
bottleneck from the last to the first: linear 1x1 (pw), depthwise 3x3 (dw), expand 1x1 (pw) -- 52 launches for 17 bottlenecks.
For each: grid, blocks per CU (256 CUs), algorithmic bytes (below), duration (median over the traced steps), the time those
bytes take at 6.3 TB/s (the streaming rate this machine sustains, profiles/r05_*), and the ratio.

Algorithmic bytes of a launch, fp32, N = 2 images (every tensor once; what an ideal kernel moves):
  pw backward  cin -> cout on hw pixels:  read  the conv's input operand (hw cin; the raw y of the block it normalises on load),
               g / D of the output (hw cout) and the output's raw y (hw cout: dy = P g + Q + R y), + D of the residual path when
               the block has one (hw cin);  write g of the input block (hw cin);  weights 2 x cin cout (read W, write dW)
  dw backward  c channels, stride s:      read  raw y1 (h w c: the operand, re-normalised on load), g2 and raw y2 (2 oh ow c);
               write g1 (h w c); weights 2 x 9 c
"""
import csv
import re
import sys
from collections import defaultdict

STAGES = ((1, 16, 1, (1,)), (2, 24, 6, (2, 1)), (3, 32, 6, (2, 1, 1)), (4, 64, 6, (2, 1, 1, 1)), (5, 96, 6, (1, 1, 1)),
          (6, 160, 6, (2, 1, 1)), (7, 320, 6, (1,)))
N, S0, C0, TAIL = 2, 256, 32, 32
HBM_TBPS = 6.3


def chain():
    """[(name, h, w, cin, wide, cout, stride, residual)] per bottleneck, forward order (512^2 image: stem output 256^2 x 32)."""
    out, h, c = [], S0, C0
    for stage, filters, t, strides in STAGES:
        for i, s in enumerate(strides):
            oh = -(-h // s)
            out.append(("bottleneck_%d_%d" % (stage, i + 1), h, h, c, c * t, filters, s, s == 1 and c == filters))
            h, c = oh, filters
    return out, h, c


def expected_launches():
    """Backward launch order: (label, kind, bytes)."""
    blocks, hl, cl = chain()
    L = [("output_conv 1x1 %d->%d @%d^2" % (cl, TAIL, hl), "pw", 4 * N * hl * hl * (cl + TAIL + cl) + 8 * cl * TAIL)]
    for name, h, w, cin, wide, cout, s, res in reversed(blocks):
        oh = -(-h // s)
        L.append(("%s linear 1x1 %d->%d @%d^2" % (name, wide, cout, oh), "pw",
                  4 * N * oh * oh * (wide + 2 * cout + wide) + 8 * wide * cout))
        L.append(("%s depthwise 3x3/%d c=%d @%d^2" % (name, s, wide, h), "dw",
                  4 * N * (h * h * wide * 2 + oh * oh * wide * 2) + 8 * 9 * wide))
        L.append(("%s expand 1x1 %d->%d @%d^2%s" % (name, cin, wide, h, " +res" if res else ""), "pw",
                  4 * N * h * h * (cin + 2 * wide + cin + (cin if res else 0)) + 8 * cin * wide))
    return L


def short(n):
    n = re.sub(r'^void\s+', '', n).replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*', '', n)


def main(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'assign_kernel' in r['Kernel_Name']]
    steps = [rows[a:b] for a, b in zip(marks, marks[1:]) if 100 < b - a < 700][-8:]
    want = expected_launches()
    per = defaultdict(list)
    meta = {}
    for st in steps:
        bw = [r for r in st if re.search(r'mb_(pw|dw)_bwd', r['Kernel_Name'])]
        if len(bw) != len(want):
            continue
        for j, r in enumerate(bw):
            kind = "dw" if "mb_dw_bwd" in r['Kernel_Name'] else "pw"
            assert kind == want[j][1], (j, r['Kernel_Name'], want[j])
            per[j].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
            blocks = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
            meta[j] = (short(r['Kernel_Name']), blocks, int(r['Workgroup_Size_X']), int(r['LDS_Block_Size']), int(r['VGPR_Count']))
    print("# MobileNetV2 chain, backward pass, cfg 2 (512^2, batch 2): %d launches per step, median duration over %d traced steps of"
          % (len(want), len(per[0])))
    print("# rocprofv3 --kernel-trace -- python bench.py --steps 10 --warmup 3 ...; bytes = algorithmic (tools/mb_bwd_layers.py docstring); "
          "floor = bytes / %.1f TB/s" % HBM_TBPS)
    print("%-3s %-52s %-48s %6s %5s %6s %5s %8s %7s %7s %6s" % ("#", "layer", "kernel", "blocks", "thr", "LDS_B", "VGPR", "MB", "us", "floor", "ratio"))
    tot_us = tot_floor = tot_b = 0.0
    fam = defaultdict(lambda: [0, 0.0, 0.0])
    for j, (label, kind, nbytes) in enumerate(want):
        d = sorted(per[j])
        us = d[len(d) // 2]
        floor = nbytes / (HBM_TBPS * 1e6)
        name, blocks, thr, lds, vgpr = meta[j]
        print("%-3d %-52s %-48s %6d %5d %6d %5d %8.2f %7.1f %7.1f %6.1f" % (j, label, name[:48], blocks, thr, lds, vgpr, nbytes / 1e6, us, floor, us / floor))
        tot_us += us; tot_floor += floor; tot_b += nbytes
        k = re.sub(r'<.*', '', name)
        fam[k][0] += 1; fam[k][1] += us; fam[k][2] += floor
    print("# total: %.1f MB, %.1f us, floor %.1f us, ratio %.1f; %.1f us per launch" % (tot_b / 1e6, tot_us, tot_floor, tot_us / tot_floor, tot_us / len(want)))
    for k, (cnt, us, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("#   %-24s %2d launches %7.1f us (floor %6.1f us, x %.1f)" % (k, cnt, us, fl, us / fl))
    # by map size: where the time sits
    by = defaultdict(lambda: [0, 0.0, 0.0])
    for j, (label, kind, nbytes) in enumerate(want):
        hw = int(re.search(r'@(\d+)\^2', label).group(1))
        d = sorted(per[j])
        by[hw][0] += 1; by[hw][1] += d[len(d) // 2]; by[hw][2] += nbytes / (HBM_TBPS * 1e6)
    for hw, (cnt, us, fl) in sorted(by.items(), reverse=True):
        print("#   maps %3d^2: %2d launches %7.1f us (floor %6.1f us, x %.1f, %.1f us per launch)" % (hw, cnt, us, fl, us / fl, us / cnt))


if __name__ == "__main__":
    main(sys.argv[1])
