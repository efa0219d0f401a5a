"""rocprofv3 --pmc target: the dominant kernel of the training step in isolation (GPU box only).
The batched product of the Winograd F(4x4,3x3) head-tower layer: 36 x [682 x 256] x [256 x 256], 30 launches.
usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/gemm_pmc.py   (WRITE_SIZE in a second pass)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow_amd"))

import torch  # noqa: E402

import _rn  # noqa: E402

dev = torch.device("cuda:0")
tiles = 2 * sum(((s + 3) // 4) ** 2 for s in (64, 32, 16, 8, 4))
A = torch.randn(36, tiles, 256, device=dev)
B = torch.randn(36, 256, 256, device=dev) * 0.01
Cm = torch.empty(36, tiles, 256, device=dev)
L = _rn.lib()
for _ in range(30):
    _rn.check(L.rn_gemm_batched(_rn.f32(A), _rn.f32(B), _rn.f32(Cm), tiles, 256, 256, 36, 0, _rn.stream()), "rn_gemm_batched")
torch.cuda.synchronize()
print("tiles", tiles, "algorithmic bytes", 4 * 36 * (2 * tiles * 256 + 256 * 256))
