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
