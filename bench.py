#!/usr/bin/env python3
"""Headline benchmark: training images/sec of MobileNetV2-FPN RetinaNet, 512x512, batch 2 per GPU
(BASELINE.json configs[1]), one process per GPU, gradients averaged with RCCL over xGMI.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = anchor assignment for the batch + forward + focal/smooth-L1 loss + backward +
gradient all-reduce + momentum optimizer, fp32, dropout 0.2 (reference default), on a synthetic
COCO-shaped batch [image, hflip(image)] that is resident in HBM before the timed region.
Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     : the dominant kernel (the batched fp32-MFMA product of the Winograd F(4x4,3x3) head-tower
                 layer, 3x3 256->256 over P3..P7), timed live with HIP events on the launch stream, against
                 the 157.3 TFLOP/s dense fp32 MFMA peak of MI355X_MICROARCH.md; executed (not direct-conv
                 equivalent) FLOPs.  The whole layer's direct-conv-equivalent rate is reported beside it;
  cpu_baseline : the CPU oracle (restatement of the reference's TF semantics, TF itself is not
                 installable) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "retinanet-tensorflow_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

IMAGE_SIZE = 512
BATCH = 2
NUM_CLASSES = 80
MAX_OBJ = 32
FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
# HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (separate --pmc FETCH_SIZE /
# --pmc WRITE_SIZE runs of tools/gemm_pmc.py, profiles/r01_gemm_pmc_fetch_write.csv):
# FETCH_SIZE 17 567 KB x 2 (gfx950 reports 1/2 of wide 16-B/lane reads, MI355X_MICROARCH.md HBM section)
# + WRITE_SIZE 24 552 KB = 61.1 MB.  Algorithmic bytes: 36 x (682x256 in + 256x256 weights + 682x256 out) x 4 = 59.7 MB.
DOMINANT_KERNEL_HBM_BYTES = (2 * 17567 + 24552) * 1024
TRAIN_GFLOP_PER_IMAGE = 239.2       # BASELINE.md section 3 (3 x forward conv FLOPs), cfg 2


def synthetic_objects(rng, image_size=IMAGE_SIZE, max_obj=MAX_OBJ):
    """O ~ clip(Poisson(7),1,32), class ~ U{0..79}, centre ~ U(0,1)^2,
    side = S*2^U(-4,-1) px with aspect 2^U(-1,1), clipped to the image."""
    o = int(np.clip(rng.poisson(7), 1, max_obj))
    cy, cx = rng.uniform(0, 1, o), rng.uniform(0, 1, o)
    side = 2.0 ** rng.uniform(-4, -1, o)
    asp = 2.0 ** rng.uniform(-1, 1, o)
    h, w = side * np.sqrt(asp), side / np.sqrt(asp)
    y1, x1 = np.clip(cy - h / 2, 0, 1), np.clip(cx - w / 2, 0, 1)
    y2 = np.minimum(np.maximum(np.clip(cy + h / 2, 0, 1), y1 + 2.0 / image_size), 1.0)
    x2 = np.minimum(np.maximum(np.clip(cx + w / 2, 0, 1), x1 + 2.0 / image_size), 1.0)
    boxes = np.zeros((max_obj, 4), np.float32)
    cls = np.zeros((max_obj,), np.int32)
    boxes[:o] = np.stack([y1, x1, y2, x2], 1)
    cls[:o] = rng.integers(0, NUM_CLASSES, o)
    return boxes, cls, o


def make_batch(rank, device):
    """[image, hflip(image)] (dataset.py:182-204) + the objects of both, on the device."""
    import dataset
    rng = np.random.default_rng(1234 + rank)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    img = torch.randn((1, IMAGE_SIZE, IMAGE_SIZE, 3), generator=g)
    image = torch.cat([img, torch.flip(img, [2])], 0).to(device).contiguous()
    boxes, cls, o = synthetic_objects(rng)
    b = torch.from_numpy(boxes)
    boxes2 = torch.stack([b, dataset.flip_boxes(b)], 0).to(device).contiguous()
    boxes2[1, o:] = 0
    cls2 = torch.from_numpy(np.stack([cls, cls])).to(device).contiguous()
    nobj = torch.tensor([o, o], dtype=torch.int32, device=device)
    return image, boxes2, cls2, nobj


class Step(object):
    """assignment + train step; the replica-local work is train.Trainer's two backward segments (each one hipGraph), the
    gradient all-reduce of the heads + FPN region runs under the backbone's backward pass."""

    def __init__(self, device, use_graph, loss_mode, dropout, rank, overlap=True, force_collective=False):
        import dataset, layers, levels, retinanet, train
        torch.manual_seed(0)                       # identical initial weights on every rank
        self.levels = levels.build_levels()
        if os.environ.get("RN_HEADS_TWO_STREAMS"):     # tuning aid (see retinanet.HEADS_TWO_STREAMS)
            retinanet.HEADS_TWO_STREAMS = os.environ["RN_HEADS_TWO_STREAMS"] == "1"
        self.net = retinanet.RetinaNet('mobilenet_v2', self.levels, NUM_CLASSES, layers.elu, dropout).to(device)
        self.image, self.boxes, self.cls, self.nobj = make_batch(rank, device)
        self.dataset = dataset
        self.trainer = train.Trainer(self.net, self.levels, optimizer='momentum', learning_rate=1e-2,
                                     loss_mode=loss_mode, device=device, use_graph=use_graph, overlap=overlap,
                                     force_collective=force_collective, input_fn=self.features,
                                     wgrad_side_stream=os.environ.get("RN_WGRAD_SIDE_STREAM") == "1")
        train.broadcast_initial_state(self.trainer)

    def features(self):
        """Device-side anchor assignment of the batch (inside the timed step, inside segment A's graph)."""
        c, r, m = self.dataset.build_labels((IMAGE_SIZE, IMAGE_SIZE), self.cls, self.boxes, self.levels, NUM_CLASSES,
                                            num_obj=self.nobj)
        return {'image': self.image, 'detection': {'classifications': c, 'regressions': r}, 'trainable_masks': m}

    def __call__(self):
        out = self.trainer.step()
        return out['class_loss'], out['regr_loss']


def time_dominant_kernel(device, iters=100):
    """The kernel the training step spends most matrix-core time in: the batched product of the Winograd
    F(4x4,3x3) head-tower layer (3x3, 256->256, the five pyramid levels of a 512^2 batch of 2 = 682 4x4 tiles):
    36 x ([682 x 256] x [256 x 256]) in ONE launch of conv_fwd_kernel<64,64,...>.  Executed FLOPs per launch =
    2 * 36 * 682 * 256 * 256 = 3.218 GFLOP (kernels table); average duration from HIP events on the
    launch stream.  Also times the whole layer (weight / input transforms + product + output transform) and reports
    its rate in direct-convolution FLOPs (12.87 GFLOP per layer)."""
    import _rn
    import ops
    sizes = [64, 32, 16, 8, 4]
    tiles = BATCH * sum(((s + 3) // 4) ** 2 for s in sizes)
    A = torch.randn(36, tiles, 256, device=device)
    B = torch.randn(36, 256, 256, device=device) * 0.01
    Cm = torch.empty(36, tiles, 256, device=device)
    L = _rn.lib()

    def timed(fn):
        # steady state: the clocks of an idle MI355X take >= 10 ms of continuous work to come up (the same kernel measures
        # 41 us in the first 50 launches after a pause and 35-36 us from then on), so warm up for 100 ms, not 5 launches
        t_end = time.perf_counter() + 0.1
        while time.perf_counter() < t_end:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters

    gemm_ms = timed(lambda: _rn.check(L.rn_gemm_batched(_rn.f32(A), _rn.f32(B), _rn.f32(Cm), tiles, 256, 256, 36, 0,
                                                        _rn.stream()), "rn_gemm_batched"))
    xs = [torch.randn(BATCH, s, s, 256, device=device) for s in sizes]
    w = torch.randn(3, 3, 256, 256, device=device) * 0.01
    with torch.no_grad():
        layer_ms = timed(lambda: ops.conv2d(xs, w, None, 1))
    pixels = BATCH * sum(s * s for s in sizes)
    return {"gemm_ms": gemm_ms, "gemm_flops": 2.0 * 36 * tiles * 256 * 256, "layer_ms": layer_ms,
            "layer_direct_flops": 2.0 * pixels * 2304 * 256,
            "gemm_bytes": 4.0 * 36 * (2 * tiles * 256 + 256 * 256)}


def nms_benchmark(device, batch=16, image_size=1024, hot=0.01, iters=5):
    """Second half of BASELINE's metric ("NMS boxes/ms"): anchor decode + candidate extraction +
    batched class-wise NMS at the shape of BASELINE configs[4] (1024x1024, batch 16, 80 classes,
    196 416 anchors per image, 3.14 M per batch), synthetic class probabilities with ~1 % of the
    anchors above the 0.5 threshold, fp32.  boxes/ms = candidates entering NMS per ms of
    candidate scan + compaction + decode of the candidates + sort + NMS; anchors/ms = rows scanned per ms."""
    import levels as levels_mod
    import utils
    lv = levels_mod.build_levels()
    g = torch.Generator(device=device).manual_seed(7)
    probs, regs = {}, {}
    size = image_size
    for i, k in enumerate(lv):
        s = -(-size // (2 ** (3 + i)))
        p = torch.rand((batch, s, s, 9, NUM_CLASSES), generator=g, device=device) * 0.45
        sel = torch.rand((batch, s, s, 9), generator=g, device=device) < hot
        cls = torch.randint(0, NUM_CLASSES, (batch, s, s, 9), generator=g, device=device)
        val = 0.5 + 0.5 * torch.rand((batch, s, s, 9), generator=g, device=device)
        p.view(-1, NUM_CLASSES)[sel.view(-1).nonzero().squeeze(1), cls.view(-1)[sel.view(-1)]] = val.view(-1)[sel.view(-1)]
        probs[k] = p
        regs[k] = torch.randn((batch, s, s, 9, 4), generator=g, device=device) * 0.3
    rows = sum(int(v.numel() // NUM_CLASSES) for v in probs.values())
    anchors = {k: lv[k].normalized_anchor_sizes((image_size, image_size)) for k in lv}

    def run():   # the raw regressions go in: only the rows that become candidates are decoded (utils.detect_raw)
        return utils.detect_raw(probs, regs, anchors, NUM_CLASSES, capacity=int(rows * 0.05), return_raw=True)

    out = run()
    torch.cuda.synchronize()
    counts = out[5].cpu().tolist()
    # steady state, like the training step: clocks up (100 ms of warm-up runs), and the ~30 launches of one batch replayed
    # as a hipGraph so that the number is device time, not the Python / ctypes launch path
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        run()
        torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        run()
    graph.replay()
    torch.cuda.synchronize()
    iters = max(iters, 20)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        graph.replay()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    read_bytes = rows * (NUM_CLASSES + 4) * 4
    return {"boxes_per_ms": round(counts[0] / ms, 1), "anchors_per_ms": round(rows / ms, 1), "ms_per_batch": round(ms, 3),
            "candidates": counts[0], "kept": counts[1], "anchors": rows, "scan_GBps": round(read_bytes / ms / 1e6, 1),
            "config": "BASELINE configs[4] shape: 1024x1024, batch %d, 80 classes, fp32 probabilities, ~1%% of anchors > 0.5; "
                      "device time of one batch (its launches replayed as a hipGraph, clocks warmed up)" % batch}


def cpu_baseline(max_seconds=30.0):
    """Oracle (torch-CPU fp32 restatement of the reference graph) on this host: forward + loss +
    backward + momentum step of the SAME workload (512^2, batch 2, 80 classes), bounded sample."""
    from oracle import dataset_ref, model_ref, train_ref
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))      # more threads than that only adds contention on these small convs
    torch.set_num_threads(cores)
    rng = np.random.default_rng(1234)
    params = model_ref.init_params("mobilenet_v2", num_classes=NUM_CLASSES, seed=0)
    # calibration: backbone forward only (~2 % of the step's FLOPs); if the host is too slow for a
    # full step inside the budget, report the forward-only rate of the bounded sample instead
    cal = torch.from_numpy(rng.standard_normal((BATCH, IMAGE_SIZE, IMAGE_SIZE, 3)).astype(np.float32))
    with torch.no_grad():
        model_ref.mobilenet_v2_forward(params, cal)
        tc = time.perf_counter()
        model_ref.mobilenet_v2_forward(params, cal)
        tc = time.perf_counter() - tc
    if tc > 4.0:
        return {"value": None, "unit": "images/sec", "cores": cores, "kind": "port",
                "sample": "skipped: backbone forward alone took %.1f s on this host (budget 30 s for the sample)" % tc}
    img = rng.standard_normal((1, IMAGE_SIZE, IMAGE_SIZE, 3)).astype(np.float32)
    image = torch.from_numpy(np.concatenate([img, img[:, :, ::-1]], 0).copy())
    boxes, cls, o = synthetic_objects(rng)
    c, r, m = dataset_ref.build_labels((IMAGE_SIZE, IMAGE_SIZE), cls[:o], boxes[:o], NUM_CLASSES)
    fc, fr, fm, _ = dataset_ref.flip(c, r, m)
    labels = {"classifications": {k: torch.from_numpy(np.stack([c[k], fc[k]])) for k in c},
              "regressions": {k: torch.from_numpy(np.stack([r[k], fr[k]])) for k in c},
              "trainable_masks": {k: torch.from_numpy(np.stack([m[k], fm[k]])) for k in c}}
    state, steps, t0 = {}, 0, time.perf_counter()
    while True:
        train_ref.train_step(params, image, labels, NUM_CLASSES, state, lr=1e-2, step=steps + 1, loss_mode="focal")
        steps += 1
        el = time.perf_counter() - t0
        if steps >= 3 or el > max_seconds / 2:
            break
    el = time.perf_counter() - t0
    return {"value": round(BATCH * steps / el, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d full train steps (fwd+focal/huber loss+bwd+momentum) of the same 512x512 batch-2 workload, "
                      "torch-CPU fp32 oracle restating the reference's TF graph (TensorFlow not installable)" % steps}


def _free_port():
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) with
    torch.distributed.run as a CHILD process and hand back its exit code.  This parent has not touched the GPU
    (torch.cuda.device_count() does not initialise it), and it never replaces itself with another program."""
    import subprocess
    have = torch.cuda.device_count()
    if args.gpus > have:
        raise SystemExit("bench.py --gpus %d: this node has %d GPU(s)" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def ensure_built():
    """(Re)build librn_hip.so when it is missing or older than a kernel source (hipcc cross-compiles; seconds when warm)."""
    import glob
    import subprocess
    src = os.path.join(ROOT, "retinanet-tensorflow_amd", "csrc")
    so = os.path.join(ROOT, "retinanet-tensorflow_amd", "librn_hip.so")
    deps = glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(src, "*.cpp")) + \
        [os.path.join(ROOT, "include", "rn_hip.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["make", "-C", src, "-j8"], check=True, stdout=sys.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying hipGraphs")
    ap.add_argument("--loss", default="focal", choices=["focal", "bce_dice"])
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nms", action="store_true", help="skip the decode+NMS throughput measurement")
    ap.add_argument("--no-roofline", action="store_true", help="skip the kernel micro-timings behind `roofline`")
    ap.add_argument("--no-overlap", action="store_true", help="one backward segment, all-reduce after it (A/B aid)")
    ap.add_argument("--spawn", action="store_true", help="go through the rank launcher even for --gpus 1")
    ap.add_argument("--force-collective", action="store_true",
                    help="issue the RCCL all-reduces even with one rank (self-test of the multi-GPU path on a 1-GPU box)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not launched and (args.gpus > 1 or args.spawn):
        ensure_built()
        sys.exit(spawn_ranks(args, [a for a in sys.argv[1:] if a != "--spawn"]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if launched and world != args.gpus:
        raise SystemExit("bench.py --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if rank == 0 and not launched:
        ensure_built()
    import train
    device, rank, world, started = train.init_distributed()
    dist = None
    if started:
        import torch.distributed as dist
        world = dist.get_world_size()              # the ranks RCCL actually connected
    import _rn
    _rn.lib()

    step = Step(device, use_graph=not args.no_graph, loss_mode=args.loss, dropout=args.dropout, rank=rank,
                overlap=not args.no_overlap, force_collective=args.force_collective)
    for _ in range(args.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step.trainer.timing = {}

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    exposed = step.trainer.allreduce_exposed_ms()
    if dist is not None:
        t = torch.tensor([elapsed, exposed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, exposed = float(t[0].item()), float(t[1].item())
    losses = [float(x) for x in out]
    step.trainer.check_device_errors()

    result = None
    if rank == 0:
        ips = world * BATCH * args.steps / elapsed
        dk = time_dominant_kernel(device)
        achieved = dk["gemm_flops"] / (dk["gemm_ms"] * 1e-3) / 1e12
        result = {
            "metric": "train images/sec (MobileNetV2-FPN RetinaNet 512x512, bs=2/GPU)",
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: MobileNetV2-FPN 512x512 bs=2/GPU, 80 classes, GroupNorm, "
                                   "%s + smooth-L1, dropout %.2f, momentum SGD, anchor assignment in the step" %
                                   (args.loss, args.dropout),
                       "global_batch": world * BATCH, "image_size": IMAGE_SIZE, "parallelism": "dp%d" % world,
                       "hip_graph": not args.no_graph, "backward_segments": 2 if step.trainer.cut_offset else 1,
                       "allreduce": {"backend": "rccl" if dist is not None else None, "ranks": world,
                                     "collectives_issued": bool(step.trainer.allreduce.active),
                                     "bytes_overlapped_with_backbone_backward": 4 * (step.trainer.arena.count - step.trainer.cut_offset),
                                     "bytes_after_backward": 4 * step.trainer.cut_offset,
                                     "allreduce_exposed_ms": round(exposed, 4)},
                       "final_class_loss": round(losses[0], 6),
                       "final_regr_loss": round(losses[1], 6),
                       "gn_barrier_timeouts": __import__("_rn").barrier_timeouts(),
                       "conv_roofline_frac_whole_step": round(ips / world * TRAIN_GFLOP_PER_IMAGE / 1e3 /
                                                              FP32_MFMA_PEAK_TFLOPS, 4)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": DOMINANT_KERNEL_HBM_BYTES,
                         "kernel": "conv_fwd_kernel<64,64,2,2,4,true>, batched: 36 x [682x256]x[256x256], the product stage of "
                                   "the Winograd F(4x4,3x3) head-tower layer (3x3 256->256 over P3..P7)",
                         "kernel_ms": round(dk["gemm_ms"], 4), "flops_per_launch": dk["gemm_flops"],
                         "algorithmic_bytes_per_launch": dk["gemm_bytes"],
                         "layer_ms": round(dk["layer_ms"], 4),
                         "layer_direct_conv_equivalent_tflops": round(dk["layer_direct_flops"] / (dk["layer_ms"] * 1e-3) / 1e12, 1)},
        }
        if not args.no_nms:
            result["nms"] = nms_benchmark(device)
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
