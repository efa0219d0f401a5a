import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/retinanet-tensorflow_amd")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda:0")
step = bench.Step(dev, False, "focal", 0.2, 0)
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    step()
torch.cuda.synchronize()
rows = [(e.key, e.count) for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[1])
for k, c in rows[:40]:
    print("%-40s %d" % (k, c))
