"""Tuning aid: time per K-iteration of conv_fwd<64,64> as a function of resident blocks per CU.
n=1, 128x128 pixels (256 M-tiles of 64), cin=256, 3x3; cout = 64*k -> 256*k tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow_amd"))
import torch
import ops
os.environ["RN_CONV_CFG"] = os.environ.get("RN_CONV_CFG", "2")
dev = torch.device("cuda:0")
for k in (1, 2, 3, 4, 6, 8):
    x = torch.randn(1, 128, 128, 256, device=dev)
    w = torch.randn(3, 3, 256, 64 * k, device=dev) * 0.01
    with torch.no_grad():
        for _ in range(3):
            ops.conv2d(x, w, None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv2d(x, w, None, 1)
        e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * 16384 * 2304 * 64 * k
    print("blocks/CU %d: %.1f us, %.1f TFLOP/s, %.0f ns per K-iter per block-slot" % (k, ms * 1e3, fl / ms / 1e9, ms * 1e6 / 72))
