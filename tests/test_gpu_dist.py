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
    assert res["losses_plain"] == res["losses_dist"], res
    assert res["bit_equal_weights"], res
    assert 0 < res["cut_offset"] < res["count"]
    # heads + FPN region first (issued under the backbone's backward pass), the backbone region after it
    assert res["launched"][0][0] >= res["cut_offset"] and res["launched"][res["buckets_per_step"] - 1][1] <= res["cut_offset"]
    # ... and with MobileNetV2's stage cut (the chain's backward pass in two parts, train.py:261-267): three slices per step,
    # tiling the arena from the top down; the identity pass-through at the cut is exact in the forward pass (same first-step
    # losses, bit for bit) and re-groups one GroupNorm's gradient sums in the backward pass (weights within 1e-6 of the range)
    sc = res["stage_cut"]
    assert sc["first_step_losses"] == sc["first_step_losses_plain"], sc
    assert sc["max_abs_diff"] <= 1e-6 * sc["scale"], sc
    lo = sc["slices"]
    assert len(lo) == 3 and lo[0] == [res["cut_offset"], res["count"]] and lo[1][1] == res["cut_offset"] and lo[2] == [0, lo[1][0]], lo
    assert 4 * lo[2][1] <= 1 << 20, lo


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
    # heads + FPN, then the chain above the C3 tap, then the rest of the backbone (MobileNetV2's stage cut is on: a collective is active)
    assert res["config"]["backward_segments"] == 3 and len(ar["slice_schedule_bytes"]) == 3
    assert ar["bytes_overlapped_with_backbone_backward"] > ar["bytes_after_backward"] > 0
    assert ar["bytes_after_backward"] <= 1 << 20, ar            # <= 1 MB left after the last backward kernel
    assert res["config"]["gn_barrier_timeouts"] == 0
    # the multi-GPU evidence fields (every rank takes part in them: exercised here with one rank so that an 8-GPU run cannot
    # be the first time this code executes)
    assert ar["rccl_ranks"] == 1 and len(ar["allreduce_exposed_ms_per_rank"]) == 1
    assert ar["allreduce_exposed_ms_max"] >= ar["allreduce_exposed_ms_mean"] >= 0.0
    assert len(ar["slices"]) == 3 and all(sl["bytes"] > 0 and sl["ms_alone"] > 0 for sl in ar["slices"])
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
        cut, count, launched = res["cut_offset"], res["count"], [tuple(x) for x in res["launched"]]
        per = len(launched) // 3
        assert per >= 2 and len(launched) == 3 * per
        for s in range(3):
            step = launched[s * per:(s + 1) * per]
            heads = [b for b in step if b[0] >= cut]
            backbone = [b for b in step if b[1] <= cut]
            assert len(heads) + len(backbone) == per                       # no bucket straddles the cut
            assert step[:len(heads)] == heads                               # heads + FPN region first
            assert sorted(heads) [0][0] == cut and max(e for _, e in heads) == count
            assert min(b for b, _ in backbone) == 0 and max(e for _, e in backbone) == cut
