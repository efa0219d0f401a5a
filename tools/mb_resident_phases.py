"""Per-phase timing of the XCD-resident section of the MobileNetV2 chain (RN_MB_RES_STAMPS=1: rank 0 of cluster 0 stamps the
100 MHz clock at the start of a phase, when its own work is done and when the cluster's barrier lets it through).
    RN_MB_RES_STAMPS=1 python tools/mb_resident_phases.py [size] [batch]"""
import os, sys
os.environ.setdefault("RN_MB_RES_STAMPS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")]
import numpy as np
import torch
import _rn, layers, levels, ops_mb, retinanet

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
net = retinanet.RetinaNet('mobilenet_v2', levels.build_levels(), 80, layers.elu, 0.2).to(dev)
bb = net.base.backbone
x = torch.randn(batch, size, size, 3, device=dev)
names = []
orig = ops_mb._count_resident
ops_mb._count_resident = lambda n: names.append(n)
for _ in range(3):
    with torch.enable_grad():
        out = bb(x.requires_grad_(True), training=True)
torch.cuda.synchronize()
sync = _rn.resident_sync(dev)
words = sync.cpu().numpy().view(np.uint32)
st = words[1024:1024 + 32 * 40].view(np.uint64).reshape(-1, 16)
nph = names[-1]
print("resident phases: %d, error word %d" % (nph, int(words[512])))
t0 = int(st[0, 0])
tot_work = tot_bar = 0.0
for p in range(nph):
    a, b, c = (int(v) for v in st[p, :3])
    nxt = int(st[p + 1, 0]) if p + 1 < nph else c
    inner = [int(v) for v in st[p, 3:] if int(v) >= a and int(v) <= b]
    print("phase %2d: start %8.2f us  work %6.2f  barrier %6.2f  (to next start %6.2f) | inside: %s" % (
        p, (a - t0) * 0.01, (b - a) * 0.01, (c - b) * 0.01, (nxt - c) * 0.01, " ".join("%.2f" % ((v - a) * 0.01) for v in inner)))
    tot_work += (b - a) * 0.01
    tot_bar += (c - b) * 0.01
print("sum of rank 0's work %.1f us, waiting at barriers %.1f us, first start -> last end %.1f us" % (tot_work, tot_bar, (int(st[nph - 1, 2]) - t0) * 0.01))
