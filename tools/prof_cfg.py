"""rocprofv3 target: a few eager training steps of one BASELINE config (usage: prof_cfg.py resnet_50 800 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import bench_configs
print(bench_configs.run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), steps=4, warmup=2, use_graph=False))
