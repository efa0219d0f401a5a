"""rocprofv3 --pmc target: the roofline kernels of bench.py in isolation (GPU box only), 30 launches each:
  fwd  batched forward products of the Winograd F(4x4,3x3) head-tower layer: 36 x [682 x 256] x [256 x 256]
  bwd  the merged backward products of the same layer (the largest in-step kernel)
  gn   the stem: direct conv 3 -> 32 (statistics in the epilogue) + its GroupNorm (+ELU+dropout) apply pass
usage (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python tools/gemm_pmc.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python tools/gemm_pmc.py
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r02_pmc_traffic.json"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow_amd"))

import torch  # noqa: E402

import _rn  # noqa: E402
import ops  # noqa: E402

dev = torch.device("cuda:0")
tiles = 2 * sum(((s + 3) // 4) ** 2 for s in (64, 32, 16, 8, 4))
A = torch.randn(36, tiles, 256, device=dev)
B = torch.randn(36, 256, 256, device=dev) * 0.01
Cm = torch.empty(36, tiles, 256, device=dev)
dM = torch.randn(36, tiles, 256, device=dev)
L = _rn.lib()
need = L.rn_winograd_bwd_products_workspace(tiles, 256, 256, 36)
ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=dev)
nsplit = C.c_int(0)
img = torch.randn(2, 512, 512, 3, device=dev)
w_stem = torch.randn(3, 3, 3, 32, device=dev) * 0.1
gamma, beta = torch.ones(32, device=dev), torch.zeros(32, device=dev)
# product mode 1: the forward product with its kernel operand pre-split in fragment order (what the training step runs: the Winograd
# kernel transform writes U that way; here rn_x3_pack_bfrag does, a kernel of another name)
frag = bool(L.rn_get_product_mode()) and bool(L.rn_x3_bfrag_ok(tiles, 256, 256))
if frag:
    Bf = torch.empty(L.rn_x3_bfrag_bytes(256, 256, 36), dtype=torch.uint8, device=dev)
    _rn.check(L.rn_x3_pack_bfrag(_rn.f32(B), Bf.data_ptr(), 256, 256, 36, 0, _rn.stream()), "rn_x3_pack_bfrag")
for _ in range(30):
    if frag:
        _rn.check(L.rn_gemm_batched_bfrag(_rn.f32(A), Bf.data_ptr(), _rn.f32(Cm), tiles, 256, 256, 36, 1, _rn.stream()), "rn_gemm_batched_bfrag")
    else:
        _rn.check(L.rn_gemm_batched(_rn.f32(A), _rn.f32(B), _rn.f32(Cm), tiles, 256, 256, 36, 0, _rn.stream()), "rn_gemm_batched")
    _rn.check(L.rn_winograd_bwd_products(_rn.f32(A), _rn.f32(B), _rn.f32(Cm), tiles, 256, 256, _rn.f32(A), _rn.f32(dM), 256, 256, 36,
                                         ws.data_ptr(), ws.numel(), C.byref(nsplit), _rn.stream()), "rn_winograd_bwd_products")
    with torch.no_grad():
        y = ops.conv2d(img, w_stem, None, 2, gn=(32, 1e-5))
        ops.group_norm_act(y, gamma, beta, groups=32, act="elu", drop_rate=0.2, seed=1)
torch.cuda.synchronize()
print("tiles", tiles, "nsplit", nsplit.value, "fragment-ordered kernel operand", frag)
