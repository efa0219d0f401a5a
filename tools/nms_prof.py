"""rocprofv3 target: the cfg-5 decode + NMS pipeline alone (hot1pct input), 20 batches.
usage: rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nms -- python tools/nms_prof.py [stress]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import bench, utils
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "hot1pct"
logits, regs, anchors = bench._cfg5_inputs(dev, 16, 1024, kind)
rows = sum(int(v.numel() // 80) for v in logits.values())
for _ in range(20):
    utils.detect_raw(logits, regs, anchors, 80, capacity=int(rows * (0.05 if kind == "hot1pct" else 1.0)), return_raw=True, logits=True)
torch.cuda.synchronize()
