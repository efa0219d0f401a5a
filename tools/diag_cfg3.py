"""Diagnosis: ResNeXt-50-FPN 800x800 batch 2 -- stem weight gradient of the product, the fp32 oracle and the fp64 oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import dataset, layers, levels as levels_mod, retinanet, train
from helpers import coco_like_objects, to_oracle_name
from oracle import backbones_ref, losses_ref, model_ref, train_ref
LEVELS = ("P3", "P4", "P5", "P6", "P7")
backbone, size, batch, classes = sys.argv[1] if len(sys.argv) > 1 else "resnet_50", int(sys.argv[2]) if len(sys.argv) > 2 else 800, 2, 80
dev = torch.device("cuda:0")
rng = np.random.default_rng(100 + size)
lv = levels_mod.build_levels()
torch.manual_seed(21)
net = retinanet.RetinaNet(backbone, lv, classes, layers.elu, 0.0)
g = torch.Generator().manual_seed(22)
with torch.no_grad():
    for name, p in net.named_parameters():
        if name.endswith("gamma"): p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
        elif name.endswith("beta"): p.copy_(0.1 * torch.randn(p.shape, generator=g))
im = rng.standard_normal((size, size, 3)).astype(np.float32)
image = torch.from_numpy(np.stack([im, im[:, ::-1].copy()]))
b, k = coco_like_objects(rng, size)
boxes = np.zeros((1, 32, 4), np.float32); cids = np.zeros((1, 32), np.int32)
boxes[0, :len(b)], cids[0, :len(b)] = b, k
leaves32 = {kk: v.detach().clone().requires_grad_(True) for kk, v in net.named_parameters()}
net.to(dev)
pc, pr, pm = dataset.build_labels((size, size), torch.from_numpy(cids).to(dev), torch.from_numpy(boxes).to(dev), lv, classes,
                                  num_obj=torch.tensor([len(b)], dtype=torch.int32, device=dev), flip_pair=True)
feats = {"image": image.to(dev), "detection": {"classifications": pc, "regressions": pr}, "trainable_masks": pm}
trainer = train.Trainer(net, lv, optimizer="momentum", learning_rate=1e-2, loss_mode="focal", device=dev)
cl, rl = trainer.forward_backward(feats)
ghip = {n: p.grad.detach().cpu().double() for n, p in net.named_parameters()}
masks = {kk: pm[kk].cpu().bool() for kk in LEVELS}

def oracle(dtype):
    leaves = {kk: v.detach().to(dtype).requires_grad_(True) for kk, v in leaves32.items()}
    params = {to_oracle_name(kk): v for kk, v in leaves.items()}
    bparams = {kk[len("base."):]: v for kk, v in leaves.items() if kk.startswith("base.backbone")}
    fe = backbones_ref.backbone_forward(backbone, bparams, image.to(dtype))
    pyr = model_ref.fpn_forward(params, fe, "elu")
    oc = {kk: model_ref.subnet_forward(params, v, "classification_subnet", 9, classes, "elu") for kk, v in pyr.items()}
    orr = {kk: model_ref.subnet_forward(params, v, "regression_subnet", 9, 4, "elu") for kk, v in pyr.items()}
    ocl, orl = losses_ref.loss(train_ref.compact({kk: pc[kk].cpu().to(dtype) for kk in LEVELS}, masks), train_ref.compact({kk: pr[kk].cpu().to(dtype) for kk in LEVELS}, masks),
                               train_ref.compact(oc, masks), train_ref.compact(orr, masks), "focal")
    names = list(leaves.keys())
    gr = dict(zip(names, torch.autograd.grad(ocl + orl, [leaves[n] for n in names])))
    return float(ocl), float(orl), {n: v.double() for n, v in gr.items()}

t0 = time.time(); c32, r32, g32 = oracle(torch.float32); t32 = time.time() - t0
t0 = time.time(); c64, r64, g64 = oracle(torch.float64); t64 = time.time() - t0
print("losses: product %.7f %.7f | fp32 oracle %.7f %.7f | fp64 oracle %.7f %.7f | oracle seconds %.1f / %.1f" % (cl.item(), rl.item(), c32, r32, c64, r64, t32, t64))
scale = max(float(v.abs().max()) for v in g64.values())
rows = []
for n in g64:
    den = max(float(g64[n].abs().max()), 1e-3 * scale)
    rows.append((float((ghip[n] - g64[n]).abs().max()) / den, float((g32[n] - g64[n]).abs().max()) / den, float(g64[n].abs().max()) / scale, n))
rows.sort(reverse=True)
for e_hip, e_32, mag, n in rows[:10]:
    print("%-60s product-vs-fp64 %.2e   fp32-oracle-vs-fp64 %.2e   |g|/scale %.2e" % (n, e_hip, e_32, mag))
