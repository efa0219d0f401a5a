"""cfg 5 (ResNeXt-50 32x4d, 1024^2, batch 16) fp16 inference, layer by layer: every distinct conv / GroupNorm-apply launch of the
backbone's bottlenecks as the product issues them (resnet.ResNeXt_Bottleneck._call_f16_folded), timed ALONE from a replayed graph, with the bytes it must move and the FLOPs it executes -> GB/s, TFLOP/s and the time
the faster of the two rooflines would allow (HBM at 4.8 TB/s sustained, fp16 MFMA at 1.0 PFLOP/s sustained on random data).
Usage: python tools/f16_layer_probe.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import layers, ops_f16

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def graph_time(fn, iters=20, reps=4):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(reps):
            fn()
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3      # us


class Norm(object):
    """what ops_f16.conv2d_norm wants of a GroupNormalization layer"""
    def __init__(self, c):
        self.groups, self.eps = 32, 1e-5
        self.gamma = torch.ones(c, device=dev); self.beta = torch.zeros(c, device=dev)
    def build(self, c, d):
        pass


def report(name, us, bytes_, flops):
    floor = max(bytes_ / 4.8e12, flops / 1.0e15) * 1e6
    print("%-58s %8.1f us  %7.0f GB/s  %6.0f TFLOP/s   floor %6.1f us  x%.2f" % (name, us, bytes_ / us / 1e3, flops / us / 1e6, floor, us / floor), flush=True)
    return us, floor


tot = [0.0, 0.0]
stages = [(256, 128, 256, 3), (128, 256, 512, 4), (64, 512, 1024, 6), (32, 1024, 2048, 3)]     # (map, mid, out, blocks) at a 1024^2 input
prev_out = 64
for si, (s, mid, out, nblk) in enumerate(stages):
    s_in = s if si == 0 else s * 2                    # the first block of stages 2..4 strides in its 3x3 conv
    for first in (True, False):
        cin = prev_out if first else out
        hin = s_in if first else s
        n_of = 1 if first else nblk - 1
        x = torch.randn(B, hin, hin, cin, device=dev).half()
        w1 = torch.randn(1, 1, cin, mid, device=dev) * 0.05
        w2 = torch.randn(3, 3, mid // 32, mid, device=dev) * 0.05
        w3 = torch.randn(1, 1, mid, out, device=dev) * 0.05
        n1, n2, n3 = Norm(mid), Norm(mid), Norm(out)
        M_in, M = B * hin * hin, B * s * s
        tag = "stage %d %s (x%d)" % (si + 1, "first" if first else "rest ", n_of)
        p1 = ops_f16.conv2d_norm(x, w1, n1, act='relu')
        u, f = report(tag + " conv1 1x1 %d->%d @%d^2" % (cin, mid, hin), graph_time(lambda: ops_f16.conv2d_norm(x, w1, n1, act='relu')),
                      2.0 * M_in * (cin + mid), 2.0 * M_in * cin * mid)
        tot[0] += n_of * u; tot[1] += n_of * f
        stride = 2 if (first and si > 0) else 1
        if ops_f16.sg_kernel_takes(p1.y.shape, w2, stride, 32):
            a1 = p1                                   # the grouped conv applies GN1 + ReLU while its input patch goes to LDS
        else:
            u, f = report(tag + " apply GN1 %d @%d^2" % (mid, hin), graph_time(lambda: p1.materialise()), 4.0 * M_in * mid, 0.0)
            tot[0] += n_of * u; tot[1] += n_of * f
            a1 = p1.materialise()
        p2 = ops_f16.conv2d_norm(a1, w2, n2, act='relu', stride=stride, groups=32)
        u, f = report(tag + " conv2 3x3 g32 %d s%d" % (mid, stride), graph_time(lambda: ops_f16.conv2d_norm(a1, w2, n2, act='relu', stride=stride, groups=32)),
                      2.0 * (M_in + M) * mid, 2.0 * M * 9 * (mid // 32) * mid)
        tot[0] += n_of * u; tot[1] += n_of * f
        p3 = ops_f16.conv2d_norm(p2, w3, n3, act='relu')
        u, f = report(tag + " conv3 1x1 %d->%d @%d^2 (GN on load)" % (mid, out, s), graph_time(lambda: ops_f16.conv2d_norm(p2, w3, n3, act='relu')),
                      2.0 * M * (mid + out), 2.0 * M * mid * out)
        tot[0] += n_of * u; tot[1] += n_of * f
        ident = x
        if first:
            wi = torch.randn(1, 1, cin, out, device=dev) * 0.05
            ni = Norm(out)
            pi = ops_f16.conv2d_norm(x, wi, ni, act=None, stride=stride)
            u, f = report(tag + " identity 1x1 %d->%d s%d" % (cin, out, stride), graph_time(lambda: ops_f16.conv2d_norm(x, wi, ni, act=None, stride=stride)),
                          2.0 * (M_in * cin / (stride * stride) + M * out), 2.0 * M * cin * out)
            tot[0] += u; tot[1] += f
            ident = pi                                # (its GroupNorm is applied inside the block's final apply pass)
        u, f = report(tag + " apply GN3 + residual + relu %d @%d^2" % (out, s), graph_time(lambda: p3.materialise(residual=ident, act_after_residual=True)),
                      6.0 * M * out, 0.0)
        tot[0] += n_of * u; tot[1] += n_of * f
        del x, a1, p1, p2, p3, ident
        torch.cuda.empty_cache()
    prev_out = out
print("backbone stages 1-4, sum of the stand-alone times: %.2f ms; sum of the floors: %.2f ms" % (tot[0] / 1e3, tot[1] / 1e3))
