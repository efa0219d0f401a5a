import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/retinanet-tensorflow_amd")
import torch, layers, levels, retinanet
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = retinanet.RetinaNet("resnet_50", levels.build_levels(), 80, layers.elu, 0.0).to(dev)
image = torch.randn(16, 1024, 1024, 3, device=dev)
layers.set_inference_dtype("f16")
with torch.no_grad():
    for _ in range(4):
        net(image, training=False)
torch.cuda.synchronize()
