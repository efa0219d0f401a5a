import os, sys
sys.path.insert(0, "/root/repo/retinanet-tensorflow_amd")
import torch, _rn
dev = torch.device("cuda:0"); L = _rn.lib()
M, K = 682, 256
for nb in (36, 72, 144):
  for cfg in ("2", "0", "1"):
    os.environ["RN_CONV_CFG"] = cfg
    A = torch.randn(nb, M, K, device=dev); B = torch.randn(nb, K, 256, device=dev) * 0.01; Cm = torch.empty(nb, M, 256, device=dev)
    fn = lambda: L.rn_gemm_batched(_rn.f32(A), _rn.f32(B), _rn.f32(Cm), M, K, 256, nb, 0, _rn.stream())
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("batch", nb, "cfg", cfg, "%.1f us  %.0f TF" % (us, 2.0 * nb * M * K * 256 / us / 1e6), flush=True)
