"""rocprofv3 --pmc target: the fp16 3x3 head-tower conv of cfg 5 alone (256 -> 256 over P3..P7 of a 1024^2 batch of 16), 10 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import ops_f16

dev = torch.device("cuda:0")
torch.manual_seed(0)
w = torch.randn(3, 3, 256, 256, device=dev) * 0.02
xs = [torch.randn(16, s, s, 256, device=dev).half() for s in (128, 64, 32, 16, 8)]
for _ in range(10):
    ops_f16.conv2d(xs, w)
torch.cuda.synchronize()
