"""rocprofv3 target: three fp32 forward passes of BASELINE configs[4]'s net (ResNeXt-50-FPN 1024^2, batch 16), no decode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import layers, levels, retinanet
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = retinanet.RetinaNet("resnet_50", levels.build_levels(), 80, layers.elu, 0.0).to(dev)
x = torch.randn(16, 1024, 1024, 3, device=dev)
with torch.no_grad():
    for _ in range(4):
        net(x, training=False)
torch.cuda.synchronize()
