"""Is the headline step bound by the host's graph launches?  Times N replays of the one-graph step three ways: (a) the enqueue loop
alone (host time per launch, the GPU still busy behind it), (b) the loop + final synchronize (what bench.py reports), (c) HIP events on
the stream around the N replays (device-side span)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")]
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
step = bench.Step(dev, use_graph=True, loss_mode="focal", dropout=0.2, rank=0)
step.trainer.check_interval = 0
for _ in range(30):
    step()
torch.cuda.synchronize()
N = 200
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(N):
        step()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue loop %.3f ms per step (host), loop + sync %.3f ms per step = %.1f images/s, device span %.3f ms per step"
          % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N, 2 * N / (t2 - t0), e0.elapsed_time(e1) / N), flush=True)
# host time of the first launches after a synchronize (empty queue: no back-pressure)
for rep in range(3):
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); step(); ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    print("host ms of 8 consecutive launches after a sync:", " ".join("%.2f" % t for t in ts), flush=True)
