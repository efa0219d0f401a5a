"""Single-rank RCCL self-test of the data-parallel code path (the multi-GPU bench itself is launched by
the driver): initialises the 'nccl' process group with world_size 1, forces the bucketed all-reduce of the
gradient arena and checks the step still matches a non-distributed step bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bench
res = []
for forced in (False, True):
    step = bench.Step(dev, use_graph=True, loss_mode="focal", dropout=0.2, rank=0)
    if forced:
        ar = step.trainer.allreduce
        ar.world = 2                       # pretend: exercises the collective calls and the 1/world scale
        orig = ar.__call__
    for _ in range(3):
        out = step()
    if forced:
        # with world "2" on one rank the all-reduce is an identity sum, scale is 0.5
        assert abs(step.trainer.allreduce() - 0.5) < 1e-12
    torch.cuda.synchronize()
    res.append([float(x) for x in out])
    print("forced_allreduce=%s losses %s" % (forced, res[-1]), flush=True)
dist.barrier()
dist.destroy_process_group()
print("dist selftest ok")
