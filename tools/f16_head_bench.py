"""fp16 3x3 head-tower conv of cfg 5 (256 -> 256 over P3..P7 of a 1024^2 batch of 16) alone: multi-level launch, per-level launches,
and the tile configurations (RN_CONV_CFG)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import ops_f16

dev = torch.device("cuda:0")
torch.manual_seed(0)
w = torch.randn(3, 3, 256, 256, device=dev) * 0.02
xs = [torch.randn(16, s, s, 256, device=dev).half() for s in (128, 64, 32, 16, 8)]
flops = 2 * sum(16 * s * s for s in (128, 64, 32, 16, 8)) * 2304 * 256


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


t = timeit(lambda: ops_f16.conv2d(xs, w))
print("multi-level launch: %.1f us  %.0f TFLOP/s" % (t, flops / t / 1e6))
t = timeit(lambda: [ops_f16.conv2d(x, w) for x in xs])
print("one launch per level: %.1f us  %.0f TFLOP/s" % (t, flops / t / 1e6))
t = timeit(lambda: ops_f16.conv2d(xs[0], w))
print("P3 alone: %.1f us  %.0f TFLOP/s" % (t, 2 * 16 * 128 * 128 * 2304 * 256 / t / 1e6))
t = timeit(lambda: ops_f16.conv2d(xs, w), iters=200)
print("multi-level launch, 200 back to back: %.1f us  %.0f TFLOP/s" % (t, flops / t / 1e6))
if len(sys.argv) > 1 and sys.argv[1] == "power":
    # is the rate set by the power / clock management rather than by the kernel?  Same launch on all-zero operands (no toggling in
    # the multipliers) and on tiny values, and the tile shapes on P3 alone
    z = [torch.zeros_like(x) for x in xs]
    t = timeit(lambda: ops_f16.conv2d(z, w), iters=200)
    print("zero activations, 200 back to back: %.1f us  %.0f TFLOP/s" % (t, flops / t / 1e6))
    wz = torch.zeros_like(w)
    t = timeit(lambda: ops_f16.conv2d(z, wz), iters=200)
    print("zero activations and weights, 200 back to back: %.1f us  %.0f TFLOP/s" % (t, flops / t / 1e6))
    t = timeit(lambda: ops_f16.conv2d(xs, w), iters=200)
    print("random again, 200 back to back: %.1f us  %.0f TFLOP/s" % (t, flops / t / 1e6))
    for cfg in ("0", "4", "5"):
        os.environ["RN_CONV_CFG"] = cfg
        t = timeit(lambda: ops_f16.conv2d(xs[0], w), iters=100)
        print("P3 alone, cfg %s: %.1f us  %.0f TFLOP/s" % (cfg, t, 2 * 16 * 128 * 128 * 2304 * 256 / t / 1e6))
        t = timeit(lambda: ops_f16.conv2d(z[0], wz), iters=100)
        print("P3 alone, cfg %s, zeros: %.1f us  %.0f TFLOP/s" % (cfg, t, 2 * 16 * 128 * 128 * 2304 * 256 / t / 1e6))
