"""Worker processes of tests/test_gpu_dist.py (each test starts fresh processes: a process group, the HIP runtime and
RCCL are initialised once per process).  Not a test module.

  nccl1 <out.json>
      one rank, backend 'nccl' (= RCCL): the Trainer of the headline model steps three times
        (a) without a process group, one backward segment, no collective  -- the plain single-GPU step
        (b) inside the RCCL group of world size 1 with the collectives FORCED, two backward segments, each a
            hipGraph, the heads + FPN all-reduce issued between them
      and the two must agree bit for bit (sum over one rank is the identity; the segment cut does not change
      the arithmetic).
  pair <rank> <world> <port> <out_dir>
      `world` processes share cuda:0 and average gradients through a gloo group (RCCL refuses two ranks on one
      device); each rank trains on its own batch -> SURVEY a29: the replicas' weights stay identical and equal
      one process that accumulates both batches' gradients.
  rccl <out_dir>       (started by `python -m torch.distributed.run --nproc-per-node R`, R = min(#GPUs, 8) >= 2)
      the production path on a multi-GPU node: one rank per GPU, backend 'nccl' (= RCCL over xGMI), hipGraph replay,
      overlap on, three steps, each rank its own batch; rank 0 also runs the one-process accumulation of all R batches.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

SIZE, CLASSES = 128, 5


def build(dev, seed_batch, use_graph, overlap, force_collective=False, dropout=0.1, size=None, capture_collectives=None):
    SIZE = size or globals()["SIZE"]
    import dataset, layers, levels as levels_mod, retinanet, train
    layers.Dropout._next_seed[0] = 0x5EED             # same dropout streams for every net built in this process
    torch.manual_seed(0)
    lv = levels_mod.build_levels()
    net = retinanet.RetinaNet('mobilenet_v2', lv, CLASSES, layers.elu, dropout).to(dev)
    rng = np.random.default_rng(seed_batch)
    g = torch.Generator().manual_seed(seed_batch)
    image = torch.randn((2, SIZE, SIZE, 3), generator=g).to(dev)
    o = 4
    y1, x1 = rng.uniform(0, 0.5, o), rng.uniform(0, 0.5, o)
    boxes = np.stack([y1, x1, y1 + rng.uniform(0.1, 0.5, o), x1 + rng.uniform(0.1, 0.5, o)], 1).astype(np.float32)
    b = torch.from_numpy(np.stack([boxes, boxes])).to(dev)
    cls = torch.from_numpy(rng.integers(0, CLASSES, (2, o)).astype(np.int32)).to(dev)

    def features():
        c, r, m = dataset.build_labels((SIZE, SIZE), cls, b, lv, CLASSES)
        return {'image': image, 'detection': {'classifications': c, 'regressions': r}, 'trainable_masks': m}

    tr = train.Trainer(net, lv, optimizer='momentum', learning_rate=1e-2, loss_mode='focal', device=dev,
                       use_graph=use_graph, overlap=overlap, force_collective=force_collective, input_fn=features,
                       capture_collectives=capture_collectives)
    return net, tr


def nccl1(out_path):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _, plain = build(dev, 11, use_graph=False, overlap=False)
    assert plain.cut_offset == 0 and not plain.allreduce.active
    for _ in range(3):
        a = plain.step()
    wa = plain.arena.weights.clone()
    la = [float(a['class_loss']), float(a['regr_loss'])]
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    os.environ["RN_STAGE_CUTS"] = "0"          # (a): the bit-for-bit comparison -- MobileNetV2's stage cut re-groups one sum, see (c)
    _, tr = build(dev, 11, use_graph=True, overlap=True, force_collective=True)
    _, ts = build(dev, 11, use_graph=True, overlap=True, force_collective=True, capture_collectives=False)
    os.environ["RN_STAGE_CUTS"] = "1"
    assert tr.cut_offset > 0 and tr.allreduce.active and tr.allreduce.world == 1 and tr._stage_bb is None
    # round 6: RCCL's all-reduce replays correctly from a captured graph here (probed with a known answer), so the step several
    # ranks run is ONE graph with the collectives as nodes and keeps the deferred tower weight gradients; `ts` is the fallback
    # (one graph per part, eager collectives between them; the deferred products forked and joined inside the first part's graph)
    res_modes = {"capturable": bool(tr.allreduce.capturable), "defer_wgrad": bool(tr.defer_wgrad),
                 "fallback_capturable": bool(ts.allreduce.capturable), "fallback_defer_wgrad": bool(ts.defer_wgrad)}
    tr.timing = {}
    ts.timing = {}
    for _ in range(3):
        b = tr.step()
        b2 = ts.step()
    torch.cuda.synchronize()
    res_modes["whole_step_graph"] = bool(tr._graphs[5])
    res_modes["fallback_whole_step_graph"] = bool(ts._graphs[5])
    res = {"bit_equal_weights": bool(torch.equal(wa, tr.arena.weights)),
           "bit_equal_weights_fallback": bool(torch.equal(wa, ts.arena.weights)),
           "losses_plain": la, "losses_dist": [float(b['class_loss']), float(b['regr_loss'])],
           "losses_fallback": [float(b2['class_loss']), float(b2['regr_loss'])],
           "cut_offset": tr.cut_offset, "heads_offset": tr.heads_offset, "count": tr.arena.count,
           "schedule": [list(x) for x in tr.schedule], "schedule_fallback": [list(x) for x in ts.schedule],
           "exposed_ms": ts.allreduce_exposed_ms(), "modes": res_modes,
           "max_abs_diff": float((wa - tr.arena.weights).abs().max())}
    ts.check_device_errors()
    del ts
    tr.check_device_errors()
    # (c) the same with MobileNetV2's stage cut: the chain's backward pass in two parts, three gradient slices per step
    # (256 px: the fused chain runs, so the cut goes through its identity pass-through; at 128 px the backbone runs layer by layer)
    _, p2 = build(dev, 11, use_graph=False, overlap=False, size=256)
    for _ in range(3):
        first_plain = p2.step()
        if _ == 0:
            fp = [float(first_plain['class_loss']), float(first_plain['regr_loss'])]
    first_plain, wa = fp, p2.arena.weights.clone()
    _, tc = build(dev, 11, use_graph=True, overlap=True, force_collective=True, size=256)
    assert tc._stage_bb is not None
    firsts = None
    for i in range(3):
        c = tc.step()
        if i == 0:
            firsts = [float(c['class_loss']), float(c['regr_loss'])]
    torch.cuda.synchronize()
    res["stage_cut"] = {"slices": [list(x) for x in tc.schedule], "first_step_losses": firsts, "first_step_losses_plain": first_plain,
                        "max_abs_diff": float((wa - tc.arena.weights).abs().max()), "scale": float(wa.abs().max())}
    tc.check_device_errors()
    dist.barrier()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


def pair(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import _rn, ops
    ops.GN_GRID_RESIDENT = False       # several processes share this GPU: no kernel may wait for co-resident peers
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, tr = build(dev, 100 + rank, use_graph=False, overlap=True, dropout=0.0)
    assert tr.allreduce.active and tr.allreduce.world == world
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "w_%d.npy" % rank), tr.arena.weights.cpu().numpy())
    dist.barrier()
    if rank == 0:
        # one process, both replicas' batches, gradients accumulated and averaged by hand
        import train
        reps = [build(dev, 100 + r, use_graph=False, overlap=False, dropout=0.0)[1] for r in range(world)]
        main = reps[0]
        for _ in range(2):
            acc = torch.zeros_like(main.arena.grads)
            for t in reps:
                t.arena.weights.copy_(main.arena.weights)
                t.forward_backward()
                acc += t.arena.grads
            main.arena.grads.copy_(acc)
            main.opt.step(1.0 / world)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "w_single.npy"), main.arena.weights.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def rccl(out_dir):
    """R ranks, one per GPU, in-place RCCL all-reduces of the arena slices under the backbone's backward pass."""
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import train
    _, tr = build(dev, 100 + rank, use_graph=True, overlap=True, dropout=0.0)
    assert tr.allreduce.active and tr.allreduce.world == world and not tr.allreduce.host_staged
    train.broadcast_initial_state(tr)
    tr.timing = {}
    steps = 3
    for _ in range(steps):
        out = tr.step()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "w_%d.npy" % rank), tr.arena.weights.cpu().numpy())
    json.dump({"rank": rank, "world": world, "schedule": [list(x) for x in tr.schedule], "cut_offset": tr.cut_offset, "count": tr.arena.count,
               "heads_offset": tr.heads_offset, "capturable": bool(tr.allreduce.capturable), "whole": bool(tr._graphs[5]),
               "defer_wgrad": bool(tr.defer_wgrad),
               "exposed_ms": tr.allreduce_exposed_ms(), "losses": [float(out['class_loss']), float(out['regr_loss'])],
               "graph": bool(tr._graphs is not None)}, open(os.path.join(out_dir, "r_%d.json" % rank), "w"))
    tr.check_device_errors()
    dist.barrier()
    if rank == 0:
        reps = [build(dev, 100 + r, use_graph=False, overlap=False, dropout=0.0)[1] for r in range(world)]
        main = reps[0]
        for _ in range(steps):
            acc = torch.zeros_like(main.arena.grads)
            for t in reps:
                t.arena.weights.copy_(main.arena.weights)
                t.forward_backward()
                acc += t.arena.grads
            main.arena.grads.copy_(acc)
            main.opt.step(1.0 / world)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "w_single.npy"), main.arena.weights.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if sys.argv[1] == "rccl":
        rccl(sys.argv[2])
    elif sys.argv[1] == "nccl1":
        nccl1(sys.argv[2])
    elif sys.argv[1] == "pair":
        pair(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        raise SystemExit("unknown mode")
