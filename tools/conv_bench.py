"""Micro-benchmark of the conv GEMM family on the head-tower shape (tuning aid, GPU box only).
usage: python tools/conv_bench.py   -- prints TFLOP/s of fwd / dgrad / wgrad per forced tile shape."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow_amd"))


def run_one():
    import torch
    import ops
    dev = torch.device("cuda:0")
    sizes = [64, 32, 16, 8, 4]
    res = {}
    for name, cin, cout in (("tower 256->256", 256, 256), ("cls out 256->720", 256, 720)):
        xs = [torch.randn(2, s, s, cin, device=dev, requires_grad=True) for s in sizes]
        w = (torch.randn(3, 3, cin, cout, device=dev) * 0.01).requires_grad_(True)
        flops = 2.0 * 2 * sum(s * s for s in sizes) * 9 * cin * cout
        ys = ops.conv2d(xs, w, None, 1)
        dys = [torch.randn_like(y) for y in ys]

        def timeit(fn, iters=20):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / iters

        with torch.no_grad():
            t_f = timeit(lambda: ops.conv2d([x.detach() for x in xs], w.detach(), None, 1))
        # backward pieces through the C ABI directly
        import ctypes as C
        import _rn
        L = _rn.lib()
        geom = _rn.ConvGeom(3, 3, 1, cin)
        xd = [x.detach() for x in xs]
        dxs = [torch.empty_like(x) for x in xd]
        segs = ops._conv_segs(xd, w.detach(), None, None, dys, dxs)
        t_d = timeit(lambda: L.rn_conv2d_dgrad(segs, len(xd), C.byref(geom), None, 0, _rn.stream()))
        need = L.rn_conv2d_wgrad_workspace(segs, len(xd), C.byref(geom))
        ws = _rn.workspace(need, dev)
        dw = torch.empty_like(w)
        t_w = timeit(lambda: L.rn_conv2d_wgrad(segs, len(xd), C.byref(geom), dw.data_ptr(), 0, ws.data_ptr(), ws.numel(),
                                               _rn.stream(), None))
        res[name] = tuple(round(flops / (t * 1e-3) / 1e12, 1) for t in (t_f, t_d, t_w)) + (round(t_f * 1e3), round(t_d * 1e3), round(t_w * 1e3))
    print(os.environ.get("RN_CONV_CFG", "auto"), res, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        run_one()
    else:
        for cfg in ("auto", "0", "1", "2", "3"):
            env = dict(os.environ)
            if cfg != "auto":
                env["RN_CONV_CFG"] = cfg
            else:
                env.pop("RN_CONV_CFG", None)
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, check=False)
