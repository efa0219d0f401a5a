"""The multi-GPU path on the 1-GPU box: RCCL process group of one rank with the collectives forced, the rank
launcher of bench.py, and two processes averaging gradients.  Each case runs in fresh processes."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    return dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))


def test_rccl_world1_forced_collectives_bit_equal_to_plain_step(tmp_path):
    out = str(tmp_path / "nccl1.json")
    r = subprocess.run([sys.executable, WORKER, "nccl1", out], env=_env(), timeout=600, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert res["losses_plain"] == res["losses_dist"] == res["losses_fallback"], res
    assert res["bit_equal_weights"] and res["bit_equal_weights_fallback"], res
    cut, heads, count = res["cut_offset"], res["heads_offset"], res["count"]
    assert 0 < cut < heads < count
    m = res["modes"]
    # the step several ranks run is the step the headline times -- one graph, the collectives nodes of it,
    # the tower weight gradients deferred; the fallback (collectives cannot be captured) is one graph per part without deferral
    assert m["capturable"] and m["whole_step_graph"] and m["defer_wgrad"], m
    assert not m["fallback_capturable"] and not m["fallback_whole_step_graph"] and m["fallback_defer_wgrad"], m
    # FPN slice when segment A ends, the subnets' slice behind their deferred products, the backbone's after its backward pass
    assert res["schedule"] == [[cut, heads], [heads, count], [0, cut]], res["schedule"]
    # (the fallback keeps the deferral too: the products are forked and joined inside the first part's graph, the subnets' slice goes
    # out behind that part -- here the only part: RN_STAGE_CUTS=0)
    assert res["schedule_fallback"] == [[cut, heads], [0, cut]] or res["schedule_fallback"] == [[cut, heads], [heads, count], [0, cut]], res["schedule_fallback"]
    assert sorted(res["schedule_fallback"]) == [[0, cut], [cut, heads], [heads, count]], res["schedule_fallback"]
    # ... and with MobileNetV2's stage cut (the chain's backward pass in two parts, train.py:261-267): three slices per step,
    # tiling the arena from the top down; the identity pass-through at the cut is exact in the forward pass (same first-step
    # losses, bit for bit) and re-groups one GroupNorm's gradient sums in the backward pass (weights within 1e-6 of the range)
    sc = res["stage_cut"]
    assert sc["first_step_losses"] == sc["first_step_losses_plain"], sc
    assert sc["max_abs_diff"] <= 1e-6 * sc["scale"], sc
    lo = sc["slices"]
    assert len(lo) == 4 and lo[0] == [cut, heads] and lo[1] == [heads, count] and lo[2][1] == cut and lo[3] == [0, lo[2][0]], lo
    assert 4 * lo[3][1] <= 1 << 20, lo


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher starts the ranks itself (here N = 1 through the same path)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--force-collective",
                        "--steps", "4", "--warmup", "3", "--no-cpu-baseline", "--no-nms", "--no-roofline", "--no-extras"],
                       env={k: v for k, v in _env().items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")},
                       timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["steps"] == 4
    ar = res["config"]["allreduce"]
    assert ar["backend"] == "rccl" and ar["ranks"] == 1 and ar["collectives_issued"]
    # FPN, the subnets (behind their deferred weight gradients), then the chain above the C3 tap, then the rest of the backbone
    # (MobileNetV2's stage cut is on: a collective is active)
    assert res["config"]["backward_segments"] == 3 and len(ar["slice_schedule_bytes"]) == 4
    assert res["config"]["whole_step_in_one_graph"] and res["config"]["collectives_captured_in_graph"] and res["config"]["tower_weight_gradients_deferred"]
    # ... and the fallback path (one graph per part, eager collectives) ran too: it is where the exposed time is measured
    assert len(ar["one_graph_per_part_eager_collectives"]["slice_schedule_bytes"]) == 4
    assert ar["bytes_overlapped_with_backbone_backward"] > ar["bytes_after_backward"] > 0
    assert ar["bytes_after_backward"] <= 1 << 20, ar            # <= 1 MB left after the last backward kernel
    assert res["config"]["gn_barrier_timeouts"] == 0
    # the multi-GPU evidence fields (every rank takes part in them: exercised here with one rank so that an 8-GPU run cannot
    # be the first time this code executes)
    assert ar["rccl_ranks"] == 1 and len(ar["allreduce_exposed_ms_per_rank"]) == 1
    assert ar["allreduce_exposed_ms_max"] >= ar["allreduce_exposed_ms_mean"] >= 0.0
    assert len(ar["slices"]) == 4 and all(sl["bytes"] > 0 and sl["ms_alone"] > 0 for sl in ar["slices"])
    assert sum(sl["bytes"] for sl in ar["slices"]) == ar["bytes_overlapped_with_backbone_backward"] + ar["bytes_after_backward"]


def test_two_replicas_equal_one_process_accumulating(tmp_path):
    """R replicas (own batch each, gradients averaged by the collective) == one process accumulating the R
    batches; replicas stay identical.  Two processes share cuda:0, gloo carries the sums (RCCL needs one GPU per rank)."""
    world, port = 2, _free_port()
    procs = [subprocess.Popen([sys.executable, WORKER, "pair", str(r), str(world), str(port), str(tmp_path)], env=_env(),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    w0, w1 = np.load(tmp_path / "w_0.npy"), np.load(tmp_path / "w_1.npy")
    ws = np.load(tmp_path / "w_single.npy")
    assert np.array_equal(w0, w1)
    scale = float(np.abs(ws).max())
    assert float(np.abs(w0 - ws).max()) <= 1e-6 * scale, float(np.abs(w0 - ws).max())


def _gpu_count():
    import torch
    return torch.cuda.device_count()          # (does not initialise the GPU in this process)


@pytest.mark.skipif(_gpu_count() < 2, reason="needs >= 2 GPUs: the in-place RCCL path with more than one rank")
def test_rccl_world_n_replicas_identical_and_equal_to_accumulation(tmp_path):
    """Self-enabling on a multi-GPU node (RCCL had never run with more than one rank on this code).
    R = min(#GPUs, 8) ranks, one per GPU, backend 'nccl' (RCCL over xGMI, in place in HBM), hipGraph replay, overlap on,
    three steps, each rank its own batch (train.py:261-267 MirroredStrategy semantics):
      * every replica ends with bit-identical weights;
      * they equal ONE process that accumulates the R batches' gradients and applies their mean, to 1e-6 of the range;
      * every step issued its buckets in the planned order: the heads + FPN slice [cut, count) first (under the backbone's
        backward pass), the backbone slice [0, cut) after it."""
    world = min(_gpu_count(), 8)
    env = {k: v for k, v in _env().items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), WORKER, "rccl", str(tmp_path)]
    r = subprocess.run(cmd, env=env, timeout=1200, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    ws = [np.load(tmp_path / ("w_%d.npy" % i)) for i in range(world)]
    for w in ws[1:]:
        assert np.array_equal(ws[0], w)
    single = np.load(tmp_path / "w_single.npy")
    scale = float(np.abs(single).max())
    assert float(np.abs(ws[0] - single).max()) <= 1e-6 * scale, float(np.abs(ws[0] - single).max())
    for i in range(world):
        res = json.load(open(tmp_path / ("r_%d.json" % i)))
        assert res["world"] == world and res["graph"]
        cut, heads, count = res["cut_offset"], res["heads_offset"], res["count"]
        sched = [tuple(x) for x in res["schedule"]]
        # the slices tile the arena from the top down; only the last one (<= 1 MB) is reduced after the last backward kernel
        if res["whole"]:
            assert res["capturable"] and res["defer_wgrad"]
            assert sched[:2] == [(cut, heads), (heads, count)], sched
            rest = sched[2:]
        else:
            assert sched[0] == (cut, count), sched
            rest = sched[1:]
        assert rest[0][1] == cut and rest[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(rest, rest[1:])), sched
