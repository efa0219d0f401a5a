"""Product kernels of a head-tower layer alone (batched 36 x [682 x 256] x [256 x 256], BASELINE configs[1]): the exact fp32
matrix-core kernels (mode 0) vs the split-bf16 kernels (mode 1), forward and merged backward products; us per launch from a
replayed hipGraph of 20 launches."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")]
import torch
import _rn

dev = torch.device("cuda:0")
L = _rn.lib()
P, M, K, N = 36, 682, 256, 256
A = torch.randn(P, M, K, device=dev); B = torch.randn(P, K, N, device=dev) * 0.05; Cc = torch.empty(P, M, N, device=dev)
need = L.rn_winograd_bwd_products_workspace(M, K, N, P)
ws = torch.empty(need // 4, device=dev)
ns = C.c_int(0)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


for mode in (0, 1, 0, 1):
    L.rn_set_product_mode(mode)
    f = timed(lambda: _rn.check(L.rn_gemm_batched(_rn.f32(A), _rn.f32(B), _rn.f32(Cc), M, K, N, P, 0, _rn.stream()), "fwd"))
    d = timed(lambda: _rn.check(L.rn_gemm_batched(_rn.f32(A), _rn.f32(B), _rn.f32(Cc), M, K, N, P, 1, _rn.stream()), "dgrad"))
    b = timed(lambda: _rn.check(L.rn_winograd_bwd_products(_rn.f32(A), _rn.f32(B), _rn.f32(Cc), M, K, N, _rn.f32(A), _rn.f32(A), K, N, P,
                                                          ws.data_ptr(), need, C.byref(ns), _rn.stream()), "bwd"))
    gf = 2.0 * P * M * K * N / 1e9
    print("mode %d: forward product %.1f us (%.1f TFLOP/s fp32-equivalent), dgrad %.1f us, backward products (dgrad + wgrad) %.1f us (%d slabs; %.1f TFLOP/s)"
          % (mode, f, gf / f * 1e3, d, b, ns.value, 2 * gf / b * 1e3))

# the forward product's kernel operand pre-split in fragment order by its producer (rn_x3_pack_bfrag once, outside the timed launches:
# inside the network the Winograd kernel transform writes it); for the data-gradient product (b_nk = 1) the same image layout is
# measured too -- it gains nothing there (its fp32 operand is k-contiguous: whole 16-byte fragments from LDS), so the network keeps Urot fp32
L.rn_set_product_mode(1)
if L.rn_x3_bfrag_ok(M, K, N):
    img0 = torch.empty(L.rn_x3_bfrag_bytes(K, N, P), dtype=torch.uint8, device=dev)
    img1 = torch.empty_like(img0)
    _rn.check(L.rn_x3_pack_bfrag(_rn.f32(B), img0.data_ptr(), K, N, P, 0, _rn.stream()), "pack")
    _rn.check(L.rn_x3_pack_bfrag(_rn.f32(B), img1.data_ptr(), K, N, P, 1, _rn.stream()), "pack")
    for _ in range(2):
        f = timed(lambda: _rn.check(L.rn_gemm_batched_bfrag(_rn.f32(A), img0.data_ptr(), _rn.f32(Cc), M, K, N, P, 1, _rn.stream()), "fwd"))
        d = timed(lambda: _rn.check(L.rn_gemm_batched_bfrag(_rn.f32(A), img1.data_ptr(), _rn.f32(Cc), M, K, N, P, 0, _rn.stream()), "dgrad"))
        print("mode 1, kernel operand pre-split in fragment order: forward product %.1f us (%.1f TFLOP/s fp32-equivalent), dgrad %.1f us"
              % (f, gf / f * 1e3, d))
