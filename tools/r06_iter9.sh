#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
cat > /tmp/t.py <<'PY'
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import dataset, layers, levels as levels_mod, retinanet, train
from data_loaders.shapes import Shapes
dev = torch.device("cuda:0")
lv = levels_mod.build_levels()
loader = Shapes(None, image_size=(800, 800), seed=0)
torch.manual_seed(0)
net = retinanet.RetinaNet('resnet_50', lv, loader.num_classes, layers.elu, 0.0).to(dev)
feed = dataset.DeviceFeed(loader, lv, scale=800, device=dev)
tr = train.Trainer(net, lv, learning_rate=1e-2, loss_mode="bce_dice", device=dev, use_graph=True, input_fn=feed)
out = []
try:
    for i in range(600):
        o = tr.step()
        if i % 50 == 0 or i == 599:
            out.append("%d:%.4f/%.4f" % (i, o["class_loss"].item(), o["regr_loss"].item()))
finally:
    feed.close()
print(" ".join(out))
PY
for cfg in "RN_GCONV_DIRECT=1 RN_X3_IM2COL=1 RN_X3_CONV1X1=1" "RN_GCONV_DIRECT=0 RN_X3_IM2COL=1 RN_X3_CONV1X1=1" "RN_GCONV_DIRECT=1 RN_X3_IM2COL=0 RN_X3_CONV1X1=1" "RN_GCONV_DIRECT=0 RN_X3_IM2COL=0 RN_X3_CONV1X1=0"; do
  echo "$cfg: $(env $cfg timeout 300 python /tmp/t.py 2>&1 | tail -1 | cut -c1-400)"
done > gpurun_out/r06_i9_train.txt 2>&1
cat gpurun_out/r06_i9_train.txt
