import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/retinanet-tensorflow_amd")
import torch, ops_f16
dev = torch.device("cuda:0")
def bench(xs, cout, k=3):
    w = torch.randn(k, k, xs[0].shape[3], cout, device=dev) * 0.05
    for _ in range(3): y = ops_f16.conv2d(xs, w, None, 1, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = ops_f16.conv2d(xs, w, None, 1, 1)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = sum(2.0 * t.numel() * k * k * xs[0].shape[3] for t in y)
    print("cfg", os.environ.get("RN_CONV_CFG", "auto"), [tuple(t.shape[1:3]) for t in xs][:2], "->", cout, ": %.0f us  %.0f TF" % (us, fl / us / 1e6), flush=True)
xs = [torch.randn(16, s, s, 256, device=dev).half() for s in (128, 64, 32, 16, 8)]
for cfg in ("auto", "0", "4", "5"):
    if cfg == "auto": os.environ.pop("RN_CONV_CFG", None)
    else: os.environ["RN_CONV_CFG"] = cfg
    bench(xs, 256)
    bench(xs, 720)
    bench([torch.randn(16, 128, 128, 512, device=dev).half()], 256, 1)
