import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/retinanet-tensorflow_amd")
import torch, ops_f16
dev = torch.device("cuda:0")
c = 256
gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)
xs = [torch.randn(16, s, s, c, device=dev).half() for s in (128, 64, 32, 16, 8)]
one = [xs[0]]
for act in (None, "elu"):
    for inp in (one, xs):
        for _ in range(6):
            ops_f16.group_norm_act(inp, gamma, beta, 32, 1e-5, act)
torch.cuda.synchronize()
