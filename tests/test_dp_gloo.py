"""Data-parallel plumbing on CPU with the gloo backend, world size 2 (the N>1 path of bench.py /
train.Trainer: flat gradient arena, bucketed all-reduce, 1/world folded into the optimizer scale).
The kernels themselves need the GPU; what is checked here is the collective logic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import train
    torch.manual_seed(0)                                   # same initial weights on every rank
    m = torch.nn.Module()
    m.a = torch.nn.Parameter(torch.randn(3, 3, 8, 16))
    m.b = torch.nn.Parameter(torch.randn(5000))
    m.c = torch.nn.Parameter(torch.randn(7))
    arena = train.ParamArena(m, torch.device("cpu"))
    ar = train.GradientAllReduce(arena, bucket_bytes=8192)  # several buckets
    assert len(ar.buckets) > 2 and ar.buckets[0][0] == 0 and ar.buckets[-1][1] == arena.count
    assert all(s % train.OPT_BLOCK == 0 for s, _ in ar.buckets)
    g = torch.Generator().manual_seed(100 + rank)           # different data per rank
    for p in (m.a, m.b, m.c):
        p.grad.copy_(torch.randn(p.shape, generator=g))
    local = arena.grads.clone()
    scale = ar()
    assert scale == 1.0 / world
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(arena.grads, sum(gathered))
    # averaged SGD step applied identically everywhere -> replicas stay bit-identical
    arena.weights.sub_(1e-2 * scale * arena.grads)
    w = [torch.zeros_like(arena.weights) for _ in range(world)]
    dist.all_gather(w, arena.weights)
    assert all(torch.equal(w[0], x) for x in w)
    np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([1]))
    dist.destroy_process_group()


def _trainer_worker(rank, world, port, out_dir):
    """The REAL train.Trainer of a MobileNetV2-FPN RetinaNet (its parameter arena, its two backward segments, its
    collective schedule and bucket arithmetic) with the device work replaced: the segments write rank-specific
    gradients instead of launching kernels, the optimizer is a plain SGD update on the arena."""
    for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import layers, levels, retinanet, train
    torch.manual_seed(0)
    lv = levels.build_levels()
    net = retinanet.RetinaNet('mobilenet_v2', lv, 3, layers.elu, 0.0)
    events = []

    class FakeKernels(train.Trainer):
        def segment_a(self, features=None):
            g = torch.Generator().manual_seed(1000 * self.steps_done + rank)
            self.arena.grads.zero_()
            self.arena.grads[self.cut_offset:].copy_(torch.randn(self.arena.count - self.cut_offset, generator=g))
            events.append("A")
            # what the real segment A plans when MobileNetV2's stage cut is installed (a collective is active): the chain's
            # backward pass in two parts, cut in front of bottleneck_4_1 (mobilenet_v2.STAGE_CUT_AFTER = bottleneck_3_3)
            self._parts = [(None, None, (stage_off[0], self.cut_offset)), (None, None, (0, stage_off[0]))]
            # ... and, with the tower weight gradients deferred, the records segment B's first part launches
            self._deferred_wgrads = ["tower weight gradients"] if self.defer_wgrad else []
            return torch.zeros(()), torch.zeros(())

        def segment_b(self, part=None, join_wgrads=True, after_wgrads=None):
            # the slices above this part's were launched before this runs; they may already have been summed
            if self._deferred_wgrads:          # the real segment B forks them in front of its first part
                self._fork_wgrads(after_wgrads)
            lo, hi = self._parts[part][2]
            g = torch.Generator().manual_seed(5000 + 1000 * self.steps_done + 100 * part + rank)
            self.arena.grads[lo:hi].copy_(torch.randn(hi - lo, generator=g))
            events.append("B%d" % part)
            return (lo, hi)

    stage_off = [0]
    tr = FakeKernels(net, lv, device='cpu', learning_rate=0.1)
    tr.allreduce.per = 1 << 18                                  # 1 MB buckets: several per region
    first_fpn = next(iter(net.base.fpn.parameters()))
    offs = {id(p): o for p, (o, _) in zip(tr.arena.params, tr.arena.offsets)}
    assert tr.cut_offset == offs[id(first_fpn)] > 0
    assert all(offs[id(p)] < tr.cut_offset for p in net.base.backbone.parameters())
    # the stage cut the real forward pass would make: the trainer holds the backbone's hook, the slice above it starts at
    # bottleneck_4_1's first parameter and is 95 % of the backbone's gradients
    import mobilenet_v2
    assert tr._stage_bb is net.base.backbone and mobilenet_v2.STAGE_CUT_AFTER == 'bottleneck_3_3'
    stage_off[0] = offs[id(next(iter(net.base.backbone.bottleneck_4_1.parameters())))]
    assert 0 < stage_off[0] < tr.cut_offset and 4 * stage_off[0] <= 1 << 20 and stage_off[0] < 0.06 * tr.cut_offset
    assert all(offs[id(p)] >= tr.cut_offset for m in (net.base.fpn, net.base.classification_subnet, net.base.regression_subnet)
               for p in m.parameters())
    # the cut hook is installed only while one of the trainer's segments runs (plain autograd users of the net never see it)
    assert net.base.backward_cut is None and tr._cut_base is net.base and tr.allreduce.active and tr.allreduce.world == world
    with tr._scoped():
        assert net.base.backward_cut is not None
    assert net.base.backward_cut is None

    def sgd(scale, advance_counter=None):
        events.append("opt")
        tr.arena.weights.sub_(tr.opt.lr * scale * tr.arena.grads)
    tr.opt.step = sgd
    orig_launch = tr.allreduce.launch

    def launch(start=0, end=None):
        events.append(("launch", start, tr.arena.count if end is None else end))
        orig_launch(start, end)
    tr.allreduce.launch = launch
    for step in range(2):
        del tr.allreduce.launched[:]
        del events[:]
        tr.step({})
        # schedule: A, all-reduce(heads + FPN), B part 0 (the chain above the C3 tap), all-reduce(its slice), B part 1, all-reduce, optimizer
        assert events == ["A", ("launch", tr.cut_offset, tr.arena.count), "B0", ("launch", stage_off[0], tr.cut_offset),
                          "B1", ("launch", 0, stage_off[0]), "opt"], events
        # buckets tile the arena exactly once, on OPT_BLOCK boundaries
        cover = sorted(tr.allreduce.launched)
        assert cover[0][0] == 0 and cover[-1][1] == tr.arena.count and len(cover) > 4
        assert all(a[1] == b[0] for a, b in zip(cover, cover[1:])) and all(s % train.OPT_BLOCK == 0 for s, _ in cover)
        # the arena now holds the sum over ranks of what each rank's segments wrote
        want = torch.zeros_like(tr.arena.grads)
        for r in range(world):
            g = torch.Generator().manual_seed(1000 * step + r)
            want[tr.cut_offset:] += torch.randn(tr.arena.count - tr.cut_offset, generator=g)
            g = torch.Generator().manual_seed(5000 + 1000 * step + r)
            want[stage_off[0]:tr.cut_offset] += torch.randn(tr.cut_offset - stage_off[0], generator=g)
            g = torch.Generator().manual_seed(5000 + 1000 * step + 100 + r)
            want[:stage_off[0]] += torch.randn(stage_off[0], generator=g)
        assert torch.allclose(tr.arena.grads, want, atol=1e-6)
        assert tr.schedule == [(tr.cut_offset, tr.arena.count), (stage_off[0], tr.cut_offset), (0, stage_off[0])]
    # Round 6: the SAME step with the head towers' weight gradients deferred (what every rank runs on the GPU when the collectives
    # are nodes of the step's graph): the FPN's slice goes out when segment A ends, the subnets' slice behind the deferred
    # products -- which overwrite it AFTER segment A -- and before the backbone parts' slices
    import ops
    heads = offs[id(next(iter(net.base.classification_subnet.parameters())))]
    assert tr.heads_offset == heads and tr.cut_offset < heads < tr.arena.count
    assert all(offs[id(p)] < heads for p in net.base.fpn.parameters())
    assert all(offs[id(p)] >= heads for m in (net.base.classification_subnet, net.base.regression_subnet) for p in m.parameters())

    def fake_wgrads(records):
        assert records == ["tower weight gradients"]
        events.append("wgrads")
        g = torch.Generator().manual_seed(9000 + 1000 * tr.steps_done + rank)
        tr.arena.grads[heads:].copy_(torch.randn(tr.arena.count - heads, generator=g))
    ops.run_deferred_wgrads = fake_wgrads
    tr.defer_wgrad = True
    for step in range(2, 4):
        del tr.allreduce.launched[:]
        del events[:]
        tr.step({})
        assert events == ["A", ("launch", tr.cut_offset, heads), "wgrads", ("launch", heads, tr.arena.count), "B0",
                          ("launch", stage_off[0], tr.cut_offset), "B1", ("launch", 0, stage_off[0]), "opt"], events
        assert tr.schedule == [(tr.cut_offset, heads), (heads, tr.arena.count), (stage_off[0], tr.cut_offset), (0, stage_off[0])]
        cover = sorted(tr.allreduce.launched)
        assert cover[0][0] == 0 and cover[-1][1] == tr.arena.count
        assert all(a[1] == b[0] for a, b in zip(cover, cover[1:])) and all(s % train.OPT_BLOCK == 0 for s, _ in cover)
        want = torch.zeros_like(tr.arena.grads)
        for r in range(world):
            g = torch.Generator().manual_seed(1000 * step + r)
            want[tr.cut_offset:] += torch.randn(tr.arena.count - tr.cut_offset, generator=g)
        want[heads:] = 0                                       # (segment A's values there were overwritten by the deferred products)
        for r in range(world):
            g = torch.Generator().manual_seed(9000 + 1000 * step + r)
            want[heads:] += torch.randn(tr.arena.count - heads, generator=g)
            g = torch.Generator().manual_seed(5000 + 1000 * step + r)
            want[stage_off[0]:tr.cut_offset] += torch.randn(tr.cut_offset - stage_off[0], generator=g)
            g = torch.Generator().manual_seed(5000 + 1000 * step + 100 + r)
            want[:stage_off[0]] += torch.randn(stage_off[0], generator=g)
        assert torch.allclose(tr.arena.grads, want, atol=1e-6)
    w = [torch.zeros_like(tr.arena.weights) for _ in range(world)]
    dist.all_gather(w, tr.arena.weights)
    assert all(torch.equal(w[0], x) for x in w)                  # replicas identical after averaged updates
    # the parameters still alias the arena (p.data / p.grad are views)
    p0 = tr.arena.params[-1]
    assert p0.data_ptr() == tr.arena.weights[tr.arena.offsets[-1][0]:].data_ptr()
    np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([1]))
    dist.destroy_process_group()


def test_real_trainer_schedule_and_arena_world2(tmp_path):
    world = 2
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok_%d.npy" % r)) for r in range(world))


def test_bucketed_gradient_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok_%d.npy" % r)) for r in range(world))


def test_single_process_allreduce_is_identity():
    sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow_amd"))
    import train
    m = torch.nn.Module()
    m.a = torch.nn.Parameter(torch.randn(10))
    arena = train.ParamArena(m, torch.device("cpu"))
    assert train.GradientAllReduce(arena)() == 1.0
