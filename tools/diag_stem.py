"""Diagnosis: the stem (7x7/2, cin 3) weight gradient at 800x800 batch 2 -- product kernel, torch-CPU fp32 and torch-CPU fp64 on the
SAME (x, dy): is the difference a defect or the conditioning of a 320 000-term sum?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import ops
from oracle import tf_ops_ref as T

dev = torch.device("cuda:0")
torch.manual_seed(0)
for kind in ("iid", "smooth"):
    x = torch.randn(2, 800, 800, 3)
    w = torch.randn(7, 7, 3, 64) * 0.1
    if kind == "iid":
        dy = torch.randn(2, 400, 400, 64)
    else:       # a GroupNorm-like gradient: zero mean per (sample, channel), smooth in space, tiny amplitude
        base = torch.randn(2, 50, 50, 64)
        dy = torch.nn.functional.interpolate(base.permute(0, 3, 1, 2), size=(400, 400), mode="bilinear").permute(0, 2, 3, 1).contiguous()
        dy = dy - dy.mean((1, 2), keepdim=True)
        dy = dy * 1e-4
    xd, wd, dyd = x.to(dev), w.to(dev).requires_grad_(True), dy.to(dev)
    y = ops.conv2d(xd, wd, None, 2)
    (gw,) = torch.autograd.grad(y, wd, dyd)
    w32 = w.clone().requires_grad_(True)
    (g32,) = torch.autograd.grad(T.conv2d_same(x, w32, 2), w32, dy)
    w64 = w.double().requires_grad_(True)
    (g64,) = torch.autograd.grad(T.conv2d_same(x.double(), w64, 2), w64, dy.double())
    s = float(g64.abs().max())
    print(kind, "max|g64| %.3e  product-vs-fp64 %.3e  cpu32-vs-fp64 %.3e  product-vs-cpu32 %.3e" % (
        s, float((gw.cpu().double() - g64).abs().max()) / s, float((g32.double() - g64).abs().max()) / s,
        float((gw.cpu() - g32).abs().max()) / s))
