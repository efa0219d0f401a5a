"""rocprofv3 target: the depthwise 3x3 kernels at the largest MobileNetV2 shapes of BASELINE configs[1] (forward with GroupNorm
rows, merged backward), 10 launches each.  usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- python tools/dw_pmc.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import ops
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
for (n, h, c, s) in ((2, 128, 144, 1), (2, 128, 144, 2), (2, 256, 96, 2), (2, 64, 192, 1)):
    x = torch.randn(n, h, h, c, device=dev, requires_grad=True)
    w = torch.randn(3, 3, c, 1, device=dev, requires_grad=True)
    for _ in range(10):
        y = ops.depthwise_conv2d(x, w, s, gn=(32, 1e-5))
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
