"""Micro-benchmark of the Winograd conv on the head-tower shape (tuning aid, GPU box only).
usage: python tools/wino_bench.py [image_size batch]  -- us per call for direct / F(2x2) / F(4x4), forward and dgrad,
per forced GEMM tile shape.  Under rocprofv3 --kernel-trace --stats the per-stage kernels show up separately."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow_amd"))


def main():
    import torch
    import _rn
    import ops
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cin = cout = 256
    dev = torch.device("cuda:0")
    sizes = [-(-size // s) for s in (8, 16, 32, 64, 128)]
    xs = [torch.randn(batch, s, s, cin, device=dev) for s in sizes]
    w = torch.randn(3, 3, cin, cout, device=dev) * 0.02
    ys = [torch.empty(batch, s, s, cout, device=dev) for s in sizes]
    dxs = [torch.empty_like(x) for x in xs]
    L = _rn.lib()
    flops = 2.0 * batch * sum(s * s for s in sizes) * 9 * cin * cout
    geom = _rn.ConvGeom(3, 3, 1, cin, 1)
    segs = ops._conv_segs(xs, w, None, ys, ys, dxs)

    def timeit(fn, iters=30):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    def wino(tile, dgrad):
        need = L.rn_conv3x3_winograd_workspace(segs, len(xs), cin, cout, tile)
        ws = _rn.workspace(need, dev)
        _rn.check(L.rn_conv3x3_winograd(segs, len(xs), cin, cout, _rn.f32(w), None, dgrad, tile, ws.data_ptr(), ws.numel(),
                                        None, None, _rn.stream()), "wino")

    dw = torch.empty_like(w)

    def wino_wgrad(tile):
        need = L.rn_conv3x3_winograd_wgrad_workspace(segs, len(xs), cin, cout, tile)
        ws = _rn.workspace(need, dev)
        _rn.check(L.rn_conv3x3_winograd_wgrad(segs, len(xs), cin, cout, _rn.f32(dw), 0, tile, ws.data_ptr(), ws.numel(),
                                              None, _rn.stream()), "wino wgrad")

    def direct_wgrad():
        need = L.rn_conv2d_wgrad_workspace(segs, len(xs), C.byref(geom))
        ws = _rn.workspace(need, dev)
        L.rn_conv2d_wgrad(segs, len(xs), C.byref(geom), _rn.f32(dw), 0, ws.data_ptr(), ws.numel(), _rn.stream(), None)

    print("wgrad direct: %.0f us" % timeit(direct_wgrad))
    for wcfg in ("", "0", "1", "2"):
        for splits in ("", "1", "2", "3", "4"):
            os.environ.pop("RN_WGRAD_CFG", None)
            os.environ.pop("RN_WGRAD_SPLITS", None)
            if wcfg:
                os.environ["RN_WGRAD_CFG"] = wcfg
            if splits:
                os.environ["RN_WGRAD_SPLITS"] = splits
            print("wgrad cfg %s splits %s: F2 %.0f us  F4 %.0f us" % (wcfg or "auto", splits or "auto", timeit(lambda: wino_wgrad(2)),
                                                                    timeit(lambda: wino_wgrad(4))), flush=True)
    os.environ.pop("RN_WGRAD_CFG", None)
    os.environ.pop("RN_WGRAD_SPLITS", None)

    for cfg in ("auto", "2"):
        if cfg == "auto":
            os.environ.pop("RN_CONV_CFG", None)
        else:
            os.environ["RN_CONV_CFG"] = cfg
        r = {
            "direct fwd": timeit(lambda: L.rn_conv2d_fwd(segs, len(xs), C.byref(geom), None, 0, _rn.stream())),
            "direct dgrad": timeit(lambda: L.rn_conv2d_dgrad(segs, len(xs), C.byref(geom), None, 0, _rn.stream())),
        }
        for tile in (2, 4):
            r["F%d fwd" % tile] = timeit(lambda: wino(tile, 0))
            r["F%d dgrad" % tile] = timeit(lambda: wino(tile, 1))
        print("cfg", cfg, {k: "%.0f us (%.0f TF eff)" % (v, flops / v / 1e6) for k, v in r.items()}, flush=True)


if __name__ == "__main__":
    main()
