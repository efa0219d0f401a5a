"""GroupNorm micro-benchmark (tuning aid, GPU box only): us per forward / backward call for typical MobileNetV2-FPN
shapes, slice-resident single-kernel path vs the three-kernel path (RN_GN_NO_SLICE=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow_amd"))

import torch  # noqa: E402

import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(2, 16, 16, 960), (2, 16, 16, 160), (2, 32, 32, 384), (2, 32, 32, 64), (2, 32, 32, 256), (2, 64, 64, 192), (2, 64, 64, 256)]


def timeit(fn, iters=100):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for mode in ("slice", "three-kernel"):
    if mode == "slice":
        os.environ.pop("RN_GN_NO_SLICE", None)
    else:
        os.environ["RN_GN_NO_SLICE"] = "1"
    for shp in shapes:
        c = shp[3]
        x = torch.randn(*shp, device=dev, requires_grad=True)
        gamma = torch.ones(c, device=dev, requires_grad=True)
        beta = torch.zeros(c, device=dev, requires_grad=True)
        dy = torch.randn(*shp, device=dev)
        with torch.no_grad():
            t_f = timeit(lambda: ops.group_norm_act(x, gamma, beta, groups=32, act="relu6"))
        y = ops.group_norm_act(x, gamma, beta, groups=32, act="relu6")

        def bwd():
            y.backward(dy, retain_graph=True)
        t_b = timeit(bwd)
        print("%-13s %-20s fwd %6.1f us   bwd %6.1f us (incl. autograd overhead)" % (mode, shp, t_f, t_b), flush=True)
