import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/retinanet-tensorflow_amd")
import torch, ops_f16
dev = torch.device("cuda:0")
def bench(shape, cout, k=1, stride=1, groups=1):
    x = torch.randn(*shape, device=dev).half()
    w = torch.randn(k, k, shape[3] // groups, cout, device=dev) * 0.05
    for _ in range(3): y = ops_f16.conv2d(x, w, None, stride, groups)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = ops_f16.conv2d(x, w, None, stride, groups)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    byt = x.numel() * 2 + y.numel() * 2
    fl = 2.0 * y.numel() * k * k * shape[3] // groups
    print(shape, "->", cout, "k", k, "s", stride, "g", groups, ": %.0f us  %.2f TB/s  %.0f TF" % (us, byt / us / 1e6, fl / us / 1e6), flush=True)
bench((16, 256, 256, 64), 256)
bench((16, 256, 256, 64), 128)
bench((16, 256, 256, 256), 128)
bench((16, 256, 256, 128), 256)
bench((16, 256, 256, 128), 128, 3, 1, 32)
bench((16, 128, 128, 512), 256)
bench((16, 128, 128, 256), 512)
bench((16, 128, 128, 256), 256, 3, 1, 32)
bench((16, 128, 128, 256), 256, 3)
bench((16, 64, 64, 1024), 512)
