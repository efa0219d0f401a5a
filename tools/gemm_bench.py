"""Batched-GEMM microbenchmark (tuning aid, GPU box only): rn_gemm_batched at the Winograd head-tower shape,
K swept to separate the per-tile fixed cost from the per-K-iteration cost, per forced tile shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow_amd"))

import torch  # noqa: E402

import _rn  # noqa: E402

dev = torch.device("cuda:0")
L = _rn.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 682
for cfg in ("auto", "0", "1", "2", "3"):
    if cfg == "auto":
        os.environ.pop("RN_CONV_CFG", None)
    else:
        os.environ["RN_CONV_CFG"] = cfg
    line = []
    for K in (256, 512, 1024):
        A = torch.randn(36, M, K, device=dev)
        B = torch.randn(36, K, 256, device=dev) * 0.01
        Cm = torch.empty(36, M, 256, device=dev)
        fn = lambda: L.rn_gemm_batched(_rn.f32(A), _rn.f32(B), _rn.f32(Cm), M, K, 256, 36, 0, _rn.stream())
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        line.append("K=%d: %.1f us (%.0f TF)" % (K, us, 2.0 * 36 * M * K * 256 / us / 1e6))
    print("cfg", cfg, "  ".join(line), flush=True)
