"""Training throughput of the other BASELINE configs on ONE GPU (cfg 3: ResNeXt-50-FPN 800^2 bs 2,
cfg 4: DenseNet-121-FPN 640^2 bs 4), eager + hipGraph.  Not the headline bench (that is bench.py)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench


def run(backbone, size, batch, steps=8, warmup=3, use_graph=True):
    torch.cuda.set_device(0)
    return bench.other_config(torch.device('cuda:0'), backbone, size, batch, steps, warmup, use_graph)


if __name__ == "__main__":
    cfgs = (("resnet_50", 800, 2), ("densenet_121", 640, 4), ("mobilenet_v2", 512, 2))
    only = sys.argv[1] if len(sys.argv) > 1 else None        # e.g. `densenet_121`: that config alone (for a rocprofv3 run)
    for cfg in cfgs:
        if only is None or cfg[0] == only:
            print(json.dumps(run(*cfg)), flush=True)
            torch.cuda.empty_cache()
