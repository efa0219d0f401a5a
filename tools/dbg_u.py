"""The Winograd kernel transform in its two output formats (csrc/winograd.hip: wino_weight_body -> fp32 U, wino_weight_frag_body ->
the forward product's pre-split fragment image): pack the fp32 U with rn_x3_pack_bfrag and count the dwords per Winograd point in
which the two images differ (0 everywhere since Wino<M>::g compiles without contraction; before: ~30 % at 20 of the 36 points)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")]
import torch, _rn
L = _rn.lib(); dev = torch.device("cuda:0")
L.rn_set_product_mode(1)
torch.manual_seed(0)
cin, cout = 64, 64
w = torch.randn(3, 3, cin, cout, device=dev) * 0.05
L.rn_set_x3_bfrag(0)
nb = L.rn_conv3x3_winograd_gn_u_bytes(cin, cout, 4)
u32 = torch.zeros(nb // 4, dtype=torch.float32, device=dev)
_rn.check(L.rn_conv3x3_winograd_gn_weights(_rn.f32(w), cin, cout, 4, u32.data_ptr(), nb, None, _rn.stream()), "w")
L.rn_set_x3_bfrag(1)
uf = torch.zeros(nb, dtype=torch.uint8, device=dev)
_rn.check(L.rn_conv3x3_winograd_gn_weights(_rn.f32(w), cin, cout, 4, uf.data_ptr(), nb, None, _rn.stream()), "w")
ref = torch.zeros(nb, dtype=torch.uint8, device=dev)
U = u32[:36 * cin * cout].reshape(36, cin, cout)
_rn.check(L.rn_x3_pack_bfrag(_rn.f32(U), ref.data_ptr(), cin, cout, 36, 0, _rn.stream()), "pack")
torch.cuda.synchronize()
n = L.rn_x3_bfrag_bytes(cin, cout, 36)
a = uf[:n].view(torch.int32).reshape(36, -1); b = ref[:n].view(torch.int32).reshape(36, -1)
d = (a != b)
print("dwords differing per xi:", d.sum(1).tolist(), "of", a.shape[1])
