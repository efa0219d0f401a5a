"""Does a replayed hipGraph run independent branches (captured on two streams) concurrently?
Two latency-bound convs (tiny grid, long K): serial = 2x, concurrent = 1x."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow_amd"))
import torch
import ops
dev = torch.device("cuda:0")
x1 = torch.randn(2, 4, 4, 256, device=dev); x2 = torch.randn(2, 4, 4, 256, device=dev)
w = torch.randn(3, 3, 256, 256, device=dev) * 0.01
side = torch.cuda.Stream()

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def serial():
    with torch.no_grad():
        for _ in range(8):
            ops.conv2d(x1, w); ops.conv2d(x2, w)

def forked():
    with torch.no_grad():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        for _ in range(8):
            ops.conv2d(x1, w)
        with torch.cuda.stream(side):
            for _ in range(8):
                ops.conv2d(x2, w)
        main.wait_stream(side)

print("eager serial  %.1f us" % timeit(serial))
print("eager forked  %.1f us" % timeit(forked))
for name, fn in (("serial", serial), ("forked", forked)):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    print("graph %s  %.1f us" % (name, timeit(g.replay)))
