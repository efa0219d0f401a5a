#!/usr/bin/env python3
"""Headline benchmark: training images/sec of MobileNetV2-FPN RetinaNet, 512x512, batch 2 per GPU
(BASELINE.json configs[1]), one process per GPU, gradients averaged with RCCL over xGMI.

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = anchor assignment for the batch + forward + focal/smooth-L1 loss + backward + gradient all-reduce (the
heads + FPN slice under the backbone's backward pass) + momentum optimizer, fp32, dropout 0.2 (reference default), on a
synthetic COCO-shaped batch [image, hflip(image)] that is resident in HBM before the timed region.
Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     : the dominant kernel INSIDE the step -- the forward products of a Winograd F(4x4,3x3) head-tower layer
                 (36 x [682x256] x [256x256], 17 launches per step; the two backward product launches of the same layer are
                 `entries[0]`) -- timed from a replayed hipGraph with HIP events on the launch stream.  The products run on
                 the bf16 matrix cores from exact three-way splits of the fp32 operands (csrc/gemm_x3.hip): `achieved` / `frac` =
                 the launch's ALGORITHMIC fp32 FLOPs against the 157.3 TFLOP/s dense fp32 matrix-core peak of MI355X_MICROARCH.md
                 (the yardstick of rounds 1-4; round 5 printed this as `fp32_equivalent`), `executed_bf16` = the six bf16
                 partial products per fp32 product the kernel issues, against the 2.5 PFLOP/s dense bf16 peak (round 5's top-level
                 figure).  `traffic` / `rocprof_kernel_ms` are replayed from profiles/ only while the kernel sources hash to what
                 was measured (tools/src_hash.py).  `entries` also holds the stem (the largest stand-alone GroupNorm);
  nms          : decode + candidate scan + hand-written segment sort + class-wise NMS at BASELINE configs[4]'s shape, fp16
                 logits / box deltas as the fp16 net writes them (sigmoid inside the scan), ~1 % hot and the stress input;
  cpu_baseline : the CPU oracle (restatement of the reference's TF semantics, TF itself is not installable) timed on
                 this host's cores on bounded samples: the cfg-2 train step, the cfg-1 (shapes 256^2) step, decode + NMS.
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
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# HBM bytes per launch of the roofline kernels, from THIS round's rocprofv3 PMC passes (separate --pmc FETCH_SIZE /
# --pmc WRITE_SIZE runs of tools/gemm_pmc.py, summarised by tools/pmc_traffic.py into this file); null when absent
import glob as _glob
PMC_TRAFFIC_FILE = (sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))) or [os.path.join(ROOT, "profiles", "none")])[-1]   # newest round's
TRAIN_GFLOP_PER_IMAGE = 239.2       # BASELINE.md section 3 (3 x forward conv FLOPs), cfg 2
# ... of which the 3x3 / stride-1 convs (heads 35.248 + FPN merges 3.020 of the 39.87 forward GMAC per image) run as
# Winograd F(4x4,3x3): 36 instead of 144 multiplies per 4x4 output tile = exactly 1/4 (every pyramid map of a 512^2 image is a
# whole number of tiles).  EXECUTED = (39.87 - 38.268) + 38.268 / 4 = 11.169 GMAC forward -> x 2 FLOP x 3 passes:
EXECUTED_TRAIN_GFLOP_PER_IMAGE = 67.0
# name prefix and grid of the forward product launch of a head-tower layer in a rocprofv3 trace (the `roofline` kernel)
X3_FWD_KERNEL = "gemm_x3_kernel<false, true, 2, 2, false>"                # fp32 kernel operand, split inside the product kernel (csrc/gemm_x3.hip)
X3_FWD_KERNEL_BFRAG = "gemm_x3_bfrag_kernel<true, 2, 2, false>"             # kernel operand pre-split in fragment order by the kernel transform (csrc/gemm_x3_bfrag.hip)
X3_FWD_GRID = lambda tiles: 36 * (-(-tiles // 128)) * 2       # noqa: E731
PRODUCT_REPS = 8                    # back-to-back launches per graph when the product kernels are timed alone (_graph_time)
FP16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense fp16 / bf16 matrix peak
INFERENCE_GFLOP_PER_IMAGE = 596.0   # cfg 5 forward, 297.98 GMAC per 1024^2 image


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
    # the objects of the SAMPLE only: the mirror image's labels are the sample's label maps flipped (augmentation.py:5-22),
    # written by the same assignment launch (dataset.build_labels(flip_pair=True))
    boxes1 = torch.from_numpy(boxes)[None].to(device).contiguous()
    cls1 = torch.from_numpy(cls)[None].to(device).contiguous()
    nobj = torch.tensor([o], dtype=torch.int32, device=device)
    return image, boxes1, cls1, nobj


class Step(object):
    """assignment + train step; the replica-local work is train.Trainer's two backward segments (each one hipGraph), the
    gradient all-reduce of the heads + FPN region runs under the backbone's backward pass."""

    def __init__(self, device, use_graph, loss_mode, dropout, rank, overlap=True, force_collective=False, capture_collectives=None):
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
                                     force_collective=force_collective, input_fn=self.features, capture_collectives=capture_collectives,
                                     wgrad_side_stream=os.environ.get("RN_WGRAD_SIDE_STREAM") == "1")
        train.broadcast_initial_state(self.trainer)

    def features(self):
        """Device-side anchor assignment of the batch (inside the timed step, inside segment A's graph): the sample is
        assigned once, its maps and their flipped copies fill the two batch slots -- the reference's [labels, flip(labels)]
        (dataset.py:182-204), bit for bit."""
        c, r, m = self.dataset.build_labels((IMAGE_SIZE, IMAGE_SIZE), self.cls, self.boxes, self.levels, NUM_CLASSES,
                                            num_obj=self.nobj, flip_pair=True)
        return {'image': self.image, 'detection': {'classifications': c, 'regressions': r}, 'trainable_masks': m}

    # the image is written before the step, only the labels are built here: the assignment may run on the label side
    # stream underneath the backbone's forward pass (train.Trainer.segment_a)
    features.concurrent = True

    def __call__(self):
        out = self.trainer.step()
        return out['class_loss'], out['regr_loss']


def _graph_time(fn, iters=100, reps=1):
    """Average device time of `fn` (launches on the current stream) from a replayed hipGraph, HIP events on the replay
    stream, clocks warmed up first (an idle MI355X needs >= 10 ms of continuous work: 100 ms of replays).  `reps` > 1 records
    that many back-to-back calls per graph: one stream, each launch waits for the one before it, so the time per launch is the
    kernel's duration + the 1.4 - 1.7 us between two nodes of one graph (profiles/r05_micro_boundary.txt) instead of + the
    ~4 us between two graph launches."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(reps):
            fn()
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (iters * reps)


def roofline_kernels(device):
    """The kernels the training step spends most time in, each timed alone from a replayed hipGraph:
      bwd  merged backward products of a head-tower layer (conv_bwd_kernel, 36 x ([682x256] x [256x256]^T data gradient +
           [256x682] x [682x256] weight-gradient partials) in ONE launch): 2 * 3.218 = 6.436 GFLOP executed -- the largest
           (kernel, grid) of the step (8 launches);
      fwd  forward products of the same layer (conv_fwd_kernel batched, 3.218 GFLOP);
      gn   the stem (the largest stand-alone GroupNorm of the step sits behind it; every GroupNorm after it is applied by its
           consumer): direct 3x3 / stride-2 conv 3 -> 32 with the statistics in its epilogue + ONE GroupNorm + ELU + dropout apply
           pass: HBM-bound, image read + y written + y read + output written."""
    import ctypes as C
    import _rn
    import ops
    sizes = [64, 32, 16, 8, 4]
    tiles = BATCH * sum(((s + 3) // 4) ** 2 for s in sizes)
    L = _rn.lib()
    A = torch.randn(36, tiles, 256, device=device)
    B = torch.randn(36, 256, 256, device=device) * 0.01
    Cm = torch.empty(36, tiles, 256, device=device)
    # product mode 1: the training step's tower layers hand the FORWARD product kernel its kernel operand U pre-split in fragment order
    # (written that way by the Winograd kernel transform, csrc/winograd.hip wino_weight_frag_body); here rn_x3_pack_bfrag builds
    # the same image from B once, outside the timed launches -- the timed kernel is the one the step runs
    bfrag = bool(L.rn_get_product_mode()) and bool(L.rn_x3_bfrag_ok(tiles, 256, 256))
    dM = torch.randn(36, tiles, 256, device=device)
    need = L.rn_winograd_bwd_products_workspace(tiles, 256, 256, 36)
    ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=device)
    nsplit = C.c_int(0)
    if bfrag:
        Bf = torch.empty(L.rn_x3_bfrag_bytes(256, 256, 36), dtype=torch.uint8, device=device)
        _rn.check(L.rn_x3_pack_bfrag(_rn.f32(B), Bf.data_ptr(), 256, 256, 36, 0, _rn.stream()), "rn_x3_pack_bfrag")
        fwd_ms = _graph_time(lambda: _rn.check(L.rn_gemm_batched_bfrag(_rn.f32(A), Bf.data_ptr(), _rn.f32(Cm), tiles, 256, 256, 36, 1,
                                                                       _rn.stream()), "rn_gemm_batched_bfrag"), reps=PRODUCT_REPS)
    else:
        fwd_ms = _graph_time(lambda: _rn.check(L.rn_gemm_batched(_rn.f32(A), _rn.f32(B), _rn.f32(Cm), tiles, 256, 256, 36, 0,
                                                                 _rn.stream()), "rn_gemm_batched"), reps=PRODUCT_REPS)
    bwd_ms = _graph_time(lambda: _rn.check(L.rn_winograd_bwd_products(_rn.f32(A), _rn.f32(B), _rn.f32(Cm), tiles, 256, 256,
                                                                      _rn.f32(A), _rn.f32(dM), 256, 256, 36, ws.data_ptr(),
                                                                      ws.numel(), C.byref(nsplit), _rn.stream()),
                                           "rn_winograd_bwd_products"), reps=PRODUCT_REPS)
    flops = 2.0 * 36 * tiles * 256 * 256
    plane = 4.0 * 36 * tiles * 256
    # algorithmic bytes: fwd reads V + U, writes M; bwd reads Vdy + Urot + V + dM, writes Mdx + the nsplit dU slabs
    fwd_bytes = 2 * plane + 4.0 * 36 * 256 * 256
    bwd_bytes = 4 * plane + 4.0 * 36 * 256 * 256 * (1 + max(nsplit.value, 1))
    img = torch.randn(BATCH, IMAGE_SIZE, IMAGE_SIZE, 3, device=device)
    w_stem = torch.randn(3, 3, 3, 32, device=device) * 0.1
    gamma, beta = torch.ones(32, device=device), torch.zeros(32, device=device)

    def stem():
        y = ops.conv2d(img, w_stem, None, 2, gn=(32, 1e-5))      # direct conv, GroupNorm statistics from its epilogue
        return ops.group_norm_act(y, gamma, beta, groups=32, act="elu", drop_rate=0.2, seed=1)

    with torch.no_grad():
        gn_ms = _graph_time(stem)
    # conv: read the image, write y; GroupNorm: read y, write the normalised tensor (1R + 1W)
    gn_bytes = 4.0 * (img.numel() + 3 * BATCH * 256 * 256 * 32)
    xs = [torch.randn(BATCH, s, s, 256, device=device) for s in sizes]
    w = torch.randn(3, 3, 256, 256, device=device) * 0.01
    with torch.no_grad():
        layer_ms = _graph_time(lambda: ops.conv2d(xs, w, None, 1))
    pixels = BATCH * sum(s * s for s in sizes)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from src_hash import kernel_source_hash
    src_sha = kernel_source_hash()
    traffic = {}
    if os.path.exists(PMC_TRAFFIC_FILE):
        traffic = json.load(open(PMC_TRAFFIC_FILE))
        # a stored measurement of ANOTHER workload (a stale file) must not be printed beside this one
        if traffic.get("_shape") != {"points": 36, "tiles": tiles, "cin": 256, "cout": 256, "batch": BATCH, "image": IMAGE_SIZE} \
                or int(traffic.get("_product_mode", 0)) != int(L.rn_get_product_mode()) \
                or traffic.get("_kernel_source_sha") != src_sha:      # ... nor one of OTHER kernel code (edited since it was measured)
            traffic = {}

    def rocprof_avg_us(kernel_prefix, blocks):
        """Average duration of (kernel, grid) in the newest committed kernel trace (profiles/r*_bench_kernel_trace_by_grid.txt): the
        dispatch's own begin -> end.  `kernel_ms` below is what HIP events see around replayed graphs of PRODUCT_REPS back-to-back
        launches: the same launch PLUS the 1.4 - 1.7 us between two dependent nodes of a graph; the two are printed side by side."""
        files = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_trace_by_grid.txt")))
        if not files:
            return None, None
        head = open(files[-1]).readline()
        if not head.startswith("# kernel_source_sha:") or head.split(":", 1)[1].strip() != src_sha:
            return None, None          # the committed trace is of other kernel code: not this run's evidence
        for line in open(files[-1]):
            name = line[:65].strip()
            cols = line[65:].split()
            if name.replace("void ", "").startswith(kernel_prefix) and len(cols) >= 3 and cols[0] == str(blocks):
                return float(cols[2]) / 1e3, os.path.basename(files[-1])
        return None, None

    def entry(name, kernel, bound, work, ms, peak, unit, bytes_, tkey, prof=None):
        ach = work / (ms * 1e-3) / (1e12 if bound == "mfma" else 1e9)
        prof_ms, prof_file = rocprof_avg_us(*prof) if prof else (None, None)
        return {"name": name, "kernel": kernel, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                "frac": round(ach / peak, 4), "kernel_ms": round(ms, 4), "algorithmic_bytes_per_launch": bytes_,
                "rocprof_kernel_ms": None if prof_ms is None else round(prof_ms, 4),
                "rocprof_source": None if prof_ms is None else "stored: " + prof_file,
                "traffic": traffic.get(tkey),
                # `traffic` is NOT counted in this run: it is the PMC measurement of the same kernel and shape kept in
                # profiles/ (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2 x FETCH + WRITE)
                "traffic_source": ("stored: " + os.path.basename(PMC_TRAFFIC_FILE)) if traffic.get(tkey) is not None else None}

    x3 = bool(L.rn_get_product_mode())
    if x3:
        # the products run on the bf16 matrix cores from exact three-way splits (csrc/gemm_x3.hip): SIX bf16 products per fp32 product.
        # Top level (`achieved` / `peak` / `frac`): the ALGORITHMIC fp32 FLOPs of the launch against the dense fp32 matrix-core
        # peak -- the scale rounds 1-4 were quoted on, comparable round over round, and not inflated by redundant work.
        # `executed_bf16`: the six bf16 partial products the kernel actually issues, against the dense bf16 peak of that instruction.
        def x3_entry(e, alg_flops, ms):
            e["flops_per_launch"] = alg_flops
            e["executed_bf16_flops_per_launch"] = 6 * alg_flops
            ex = 6 * alg_flops / (ms * 1e-3) / 1e12
            e["executed_bf16"] = {"achieved": round(ex, 2), "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ex / FP16_MFMA_PEAK_TFLOPS, 4),
                                  "what": "6 bf16 MFMA products per fp32 product (exact 3-way split), priced against the dense bf16 peak"}
            # timing method (changed in round 5; kept since): HIP events around replayed graphs of `reps` back-to-back launches
            e["reps"] = PRODUCT_REPS
            return e

        fwd = x3_entry(entry("head-tower layer, forward products (largest in-step kernel: 17 launches per step)",
                             ("gemm_x3_bfrag_kernel" if bfrag else "gemm_x3_kernel") + ": 36 x [682x256]x[256x256], fp32 operands split into 3 bf16, 6 bf16 MFMA products, fp32 accumulate",
                             "mfma", flops, fwd_ms, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s", fwd_bytes, "fwd_products",
                             prof=(X3_FWD_KERNEL_BFRAG if bfrag else X3_FWD_KERNEL, X3_FWD_GRID(tiles))), flops, fwd_ms)
        fwd["kernel_operand"] = ("pre-split into 3 bf16 planes in MFMA-fragment order by the Winograd kernel transform (6 B / element, read straight "
                                 "from global memory)") if bfrag else "fp32, split inside the product kernel"
        bwd = x3_entry(entry("head-tower layer, backward products: data-gradient launch + weight-gradient launch (8 pairs per step)",
                             "gemm_x3_kernel (dgrad) + gemm_x3_kernel (wgrad): 36 x ([682x256]x[256x256]^T + [256x682]x[682x256] in %d ranges)" % max(nsplit.value, 1),
                             "mfma", 2 * flops, bwd_ms, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s", bwd_bytes, "bwd_products"), 2 * flops, bwd_ms)
    else:
        bwd = entry("head-tower layer, merged backward products (largest in-step kernel, 8 launches per step)",
                    "conv_bwd_kernel<64,64,2,2,true,64,64,2,2>: 36 x ([682x256]x[256x256]^T dgrad + [256x682]x[682x256] wgrad partials)",
                    "mfma", 2 * flops, bwd_ms, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s", bwd_bytes, "bwd_products",
                    prof=("conv_bwd_kernel<64, 64, 2, 2, true, 64, 64, 2, 2>", 36 * (44 + 32)))
        bwd["flops_per_launch"] = 2 * flops
        fwd = entry("head-tower layer, forward products", "conv_fwd_kernel<64,64,2,2,4,true>, batched: 36 x [682x256]x[256x256]",
                    "mfma", flops, fwd_ms, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s", fwd_bytes, "fwd_products",
                    prof=("conv_fwd_kernel<64, 64, 2, 2, 4, true>", 36 * 44))
        fwd["flops_per_launch"] = flops
    fwd["layer_ms"] = round(layer_ms, 4)
    fwd["layer_direct_conv_equivalent_tflops"] = round(2.0 * pixels * 2304 * 256 / (layer_ms * 1e-3) / 1e12, 1)
    gn = entry("the stem: direct 3x3/2 conv 3->32 of the 512^2 batch (statistics in its epilogue) + its GroupNorm + ELU + dropout as one apply pass",
               "stem_conv_fwd_kernel + gn_apply_rows_kernel", "hbm", gn_bytes, gn_ms, HBM_PEAK_GBPS, "GB/s", gn_bytes, "group_norm")
    return (fwd, bwd, gn) if x3 else (bwd, fwd, gn)      # (first = the `roofline` object: the step's dominant kernel)


def _cfg5_inputs(device, batch, image_size, kind, seed=7):
    """fp16 class LOGITS and box deltas in the layout the fp16 net writes them (BASELINE configs[4]).
    hot1pct: ~1 % of the anchors carry one class above 0 (p > 0.5); stress: logits ~ N(-2, 2^2) i.i.d.."""
    import levels as levels_mod
    lv = levels_mod.build_levels()
    g = torch.Generator(device=device).manual_seed(seed)
    logits, regs = {}, {}
    for i, k in enumerate(lv):
        s = -(-image_size // (2 ** (3 + i)))
        shape = (batch, s, s, 9, NUM_CLASSES)
        if kind == "stress":
            z = torch.randn(shape, generator=g, device=device) * 2 - 2
        else:
            z = -1.0 - 3.0 * torch.rand(shape, generator=g, device=device)
            sel = torch.rand(shape[:-1], generator=g, device=device) < 0.01
            cls = torch.randint(0, NUM_CLASSES, shape[:-1], generator=g, device=device)
            val = 4.0 * torch.rand(shape[:-1], generator=g, device=device) + 0.01
            z.view(-1, NUM_CLASSES)[sel.view(-1).nonzero().squeeze(1), cls.view(-1)[sel.view(-1)]] = val.view(-1)[sel.view(-1)]
        logits[k] = z.half()
        regs[k] = (torch.randn((batch, s, s, 9, 4), generator=g, device=device) * 0.3).half()
    anchors = {k: lv[k].normalized_anchor_sizes((image_size, image_size)) for k in lv}
    return logits, regs, anchors


def nms_benchmark(device, batch=16, image_size=1024):
    """Second half of BASELINE's metric ("NMS boxes/ms"): candidate scan (sigmoid inside) + compaction + decode of the
    candidates + segment sort + batched class-wise NMS at BASELINE configs[4]'s shape (1024x1024, batch 16, 80 classes,
    196 416 anchors per image, 3.14 M per batch), inputs in fp16 as the fp16 net writes them: 84 x 2 bytes per anchor.
    boxes/ms = candidates entering NMS per ms of the whole pipeline; anchors/ms = rows scanned per ms.  Device time of
    one batch: its launches replayed as a hipGraph, clocks warmed up."""
    import utils
    out = {}
    for kind in ("hot1pct", "stress"):
        logits, regs, anchors = _cfg5_inputs(device, batch, image_size, kind)
        rows = sum(int(v.numel() // NUM_CLASSES) for v in logits.values())
        cap = int(rows * (0.05 if kind == "hot1pct" else 1.0))

        def run():
            return utils.detect_raw(logits, regs, anchors, NUM_CLASSES, capacity=cap, return_raw=True, logits=True)

        res = run()
        torch.cuda.synchronize()
        counts = res[5].cpu().tolist()
        ms = _graph_time(run, iters=20)
        read_bytes = rows * (NUM_CLASSES + 4) * 2
        out[kind] = {"boxes_per_ms": round(counts[0] / ms, 1), "anchors_per_ms": round(rows / ms, 1), "ms_per_batch": round(ms, 3),
                     "candidates": counts[0], "kept": counts[1], "anchors": rows,
                     "algorithmic_read_GBps": round(read_bytes / ms / 1e6, 1)}
        del logits, regs, res
    out["config"] = ("BASELINE configs[4] shape: 1024x1024, batch %d, 80 classes, fp16 logits + fp16 box deltas (sigmoid inside "
                     "the scan); hot1pct: ~1 %% of the anchors above 0.5, stress: logits ~ N(-2, 2^2); device time of one batch" % batch)
    # the headline figure of the metric string is the ~1 % hot case (what a trained detector produces)
    out.update({k: out["hot1pct"][k] for k in ("boxes_per_ms", "anchors_per_ms", "ms_per_batch")})
    return out


def _oracle_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))      # more threads than that only adds contention on these small convs
    torch.set_num_threads(cores)
    return cores


def _oracle_step_rate(image_size, classes, boxes, cls, rng, max_steps, budget_s):
    from oracle import dataset_ref, model_ref, train_ref
    params = model_ref.init_params("mobilenet_v2", num_classes=classes, seed=0)
    img = rng.standard_normal((1, image_size, image_size, 3)).astype(np.float32)
    image = torch.from_numpy(np.concatenate([img, img[:, :, ::-1]], 0).copy())
    c, r, m = dataset_ref.build_labels((image_size, image_size), cls, boxes, classes)
    fc, fr, fm, _ = dataset_ref.flip(c, r, m)
    labels = {"classifications": {k: torch.from_numpy(np.stack([c[k], fc[k]])) for k in c},
              "regressions": {k: torch.from_numpy(np.stack([r[k], fr[k]])) for k in c},
              "trainable_masks": {k: torch.from_numpy(np.stack([m[k], fm[k]])) for k in c}}
    state, steps, t0 = {}, 0, time.perf_counter()
    while True:
        train_ref.train_step(params, image, labels, classes, state, lr=1e-2, step=steps + 1, loss_mode="focal")
        steps += 1
        if steps >= max_steps or time.perf_counter() - t0 > budget_s:
            break
    return 2 * steps / (time.perf_counter() - t0), steps


def cpu_baseline(max_seconds=30.0):
    """Oracle (torch-CPU fp32 restatement of the reference graph; numpy decode + NMS) on this host, bounded samples:
      value : forward + loss + backward + momentum step of the SAME cfg-2 workload (512^2, batch 2, 80 classes)
      cfg1  : the same step at BASELINE configs[0] (shapes-style 256^2, 3 classes, [image, hflip])
      nms   : decode + class-wise NMS of one 1024^2 image of the cfg-5 ~1 % hot input (utils_ref.detect_image)."""
    from oracle import model_ref, utils_ref
    cores = _oracle_threads()
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
    boxes, cls, o = synthetic_objects(rng)
    rate, steps = _oracle_step_rate(IMAGE_SIZE, NUM_CLASSES, boxes[:o], cls[:o], rng, 3, max_seconds / 2)
    res = {"value": round(rate, 4), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": "%d full train steps (fwd+focal/huber loss+bwd+momentum) of the same 512x512 batch-2 workload, "
                     "torch-CPU fp32 oracle restating the reference's TF graph (TensorFlow not installable)" % steps}
    # cfg 1: shapes-style squares (data_loaders/shapes.py:143-176), 3 classes, 256x256
    nsq = int(rng.integers(1, 5))
    half = rng.integers(20, 64, nsq)
    ctr = rng.integers(64, 192, (nsq, 2))
    sb = np.stack([ctr[:, 0] - half, ctr[:, 1] - half, ctr[:, 0] + half, ctr[:, 1] + half], 1).astype(np.float32) / 256.0
    rate1, steps1 = _oracle_step_rate(256, 3, sb, rng.integers(0, 3, nsq).astype(np.int32), rng, 5, 6.0)
    res["cfg1"] = {"value": round(rate1, 3), "unit": "images/sec",
                   "sample": "%d train steps, BASELINE configs[0]: shapes-style 256x256, 3 classes, [image, hflip]" % steps1}
    # decode + NMS: one image of the cfg-5 ~1 % hot input, numpy oracle (single thread: the greedy loop is sequential)
    prng = np.random.default_rng(8)
    probs, regs = {}, {}
    for i, k in enumerate(("P3", "P4", "P5", "P6", "P7")):
        s = -(-1024 // 2 ** (3 + i))
        p = prng.uniform(0, 0.45, (s, s, 9, NUM_CLASSES)).astype(np.float32)
        sel = prng.uniform(size=(s, s, 9)) < 0.01
        p[sel, prng.integers(0, NUM_CLASSES, int(sel.sum()))] = prng.uniform(0.5, 1.0, int(sel.sum())).astype(np.float32)
        probs[k] = p
        regs[k] = (prng.standard_normal((s, s, 9, 4)) * 0.3).astype(np.float32)
    t0 = time.perf_counter()
    det = utils_ref.detect_image(probs, regs, (1024, 1024), NUM_CLASSES)
    el = time.perf_counter() - t0
    ncand = int(sum(int((probs[k].max(-1) > 0.5).sum()) for k in probs))
    res["nms"] = {"boxes_per_ms": round(ncand / (el * 1e3), 2), "anchors_per_ms": round(196416 / (el * 1e3), 1), "cores": 1,
                  "sample": "one 1024x1024 image (196 416 anchors, %d candidates, %d kept): numpy decode + class-wise greedy NMS"
                            % (ncand, len(det.scores))}
    return res


class SyntheticCoco(object):
    """Loader protocol (data_loaders/base.py) over synthetic COCO-shaped samples at the cfg-2 size: a pool of
    `pool` uint8 images rendered once (host RNG at 512^2 would otherwise be the bottleneck of a 4 ms step), a NEW
    (image, objects) pair every step -- the image cycles through the pool, the objects are drawn fresh."""
    class_names = ["c%d" % i for i in range(NUM_CLASSES)]
    num_classes = NUM_CLASSES

    def __init__(self, seed=0, pool=32, image_size=IMAGE_SIZE):
        self.rng = np.random.default_rng(seed)
        self.size = image_size
        self.images = [self.rng.integers(0, 256, (image_size, image_size, 3), dtype=np.uint8) for _ in range(pool)]

    def __iter__(self):
        i = 0
        while True:
            boxes, cls, o = synthetic_objects(self.rng, self.size)
            yield {"image": self.images[i % len(self.images)], "class_ids": cls[:o], "boxes": boxes[:o] * self.size}
            i += 1


def _feed_rate(device, loader, scale, num_classes, steps, warmup, loss_mode="focal", dropout=0.2):
    """images/sec of the product's training loop on FRESH data: dataset.DeviceFeed (loader thread -> pinned host memory ->
    asynchronous upload -> static device buffers) + the hipGraph step that rescales / normalises / flips / assigns inside
    its captured segment (train.py main() runs exactly this)."""
    import dataset, layers, levels, retinanet, train
    lv = levels.build_levels()
    torch.manual_seed(0)
    net = retinanet.RetinaNet('mobilenet_v2', lv, num_classes, layers.elu, dropout).to(device)
    feed = dataset.DeviceFeed(loader, lv, scale=scale, device=device)
    tr = train.Trainer(net, lv, optimizer='momentum', learning_rate=1e-2, loss_mode=loss_mode, device=device, use_graph=True,
                       input_fn=feed)
    try:
        for _ in range(warmup):
            tr.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step()
        torch.cuda.synchronize()
        return 2 * steps / (time.perf_counter() - t0)
    finally:
        feed.close()


def fresh_data(device, resident_ips, steps=100, warmup=10):
    """cfg 2 with a new image + objects every step (the reference trains on a new sample every step: train.py:190-202,
    dataset.py:182-204), next to the resident-batch headline."""
    ips = _feed_rate(device, SyntheticCoco(seed=99), None, NUM_CLASSES, steps, warmup)
    return {"value": round(ips, 1), "unit": "images/sec", "fraction_of_resident_batch": round(ips / resident_ips, 4),
            "sample": "%d hipGraph steps, each on a NEW 512x512 uint8 image (pool of 32 host images) + freshly drawn objects: loader thread -> "
                      "pinned memory -> async H2D -> uint8 -> fp32 normalise + h-flip + anchor assignment inside the captured step" % steps}


def cfg1_gpu(device, steps=100):
    """BASELINE configs[0] through the product path: the shapes loader -> DeviceFeed -> hipGraph train step at 256x256."""
    from data_loaders.shapes import Shapes
    ips = _feed_rate(device, Shapes(None, image_size=(320, 256)), 256, 3, steps, 10)
    return {"value": round(ips, 1), "unit": "images/sec",
            "sample": "%d hipGraph steps incl. the host shapes loader (background thread), upload, rescale 320x256 -> 256, label assignment: "
                      "every step is a new sample" % steps}


def other_config(device, backbone, size, batch, steps=8, warmup=3, use_graph=True):
    """Per-GPU workload of BASELINE configs[2] / configs[3] (ResNeXt-50-FPN 800^2 bs 2, DenseNet-121-FPN 640^2 bs 4) on ONE GPU:
    the same step as the headline (assignment + forward + focal / smooth-L1 + backward in stage parts + momentum)."""
    import dataset, layers, levels, retinanet, train
    torch.manual_seed(0)
    lv = levels.build_levels()
    net = retinanet.RetinaNet(backbone, lv, NUM_CLASSES, layers.elu, 0.2).to(device)
    rng = np.random.default_rng(0)
    image = torch.randn(batch, size, size, 3, device=device)
    boxes = np.zeros((batch, MAX_OBJ, 4), np.float32); cls = np.zeros((batch, MAX_OBJ), np.int32); nobj = np.zeros(batch, np.int32)
    for i in range(batch):
        boxes[i], cls[i], nobj[i] = synthetic_objects(rng, size)
    boxes, cls, nobj = (torch.from_numpy(a).to(device) for a in (boxes, cls, nobj))

    def features():
        c, r, m = dataset.build_labels((size, size), cls, boxes, lv, NUM_CLASSES, num_obj=nobj)
        return {"image": image, "detection": {"classifications": c, "regressions": r}, "trainable_masks": m}

    trainer = train.Trainer(net, lv, loss_mode="focal", device=device, use_graph=use_graph, input_fn=features)
    for _ in range(warmup):
        trainer.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    trainer.check_device_errors()
    # the gradient all-reduce schedule a multi-GPU run follows: heads + FPN after segment A, then one slice per backbone
    # part (last stage first); only the LAST part's slice is reduced after the last backward kernel
    ranges = list(trainer._graphs[2]) if (use_graph and trainer._graphs) else []
    total = 4 * trainer.arena.count
    after = 4 * (ranges[-1][1] - ranges[-1][0]) if ranges else 4 * trainer.cut_offset
    return {"backbone": backbone, "image_size": size, "batch": batch, "images_per_sec": round(batch * steps / el, 2),
            "ms_per_step": round(1e3 * el / steps, 2), "hip_graph": use_graph, "steps": steps,
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "backward_parts": 1 + len(ranges), "gradient_bytes": total, "bytes_after_backward": after,
            "bytes_after_backward_frac": round(after / total, 4),
            "allreduce_slices_MB": [round(4 * (trainer.arena.count - trainer.cut_offset) / 1e6, 1)] + [round(4 * (hi - lo) / 1e6, 1) for lo, hi in ranges]}


def inference_benchmark(device, size=1024, batch=16, iters=5):
    """BASELINE configs[4]: ResNeXt-50-FPN 1024x1024, batch 16, forward (training=False) + sigmoid (inside the candidate scan) + anchor
    decode + batched class-wise NMS over ALL 3.14 M anchors of the batch (random-init net: every anchor passes the 0.0105 threshold,
    the NMS stress case), in fp16 storage (f16 matrix-core convs, fp32 accumulate; the headline of the config) and in fp32."""
    import layers, levels, retinanet, utils
    torch.manual_seed(0)
    lv = levels.build_levels()
    net = retinanet.RetinaNet('resnet_50', lv, NUM_CLASSES, layers.elu, 0.0).to(device)
    image = torch.randn(batch, size, size, 3, device=device)
    anchors = {k: lv[k].normalized_anchor_sizes((size, size)) for k in lv}

    def run():
        with torch.no_grad():
            out = net(image, training=False)
            rows = sum(v.numel() // NUM_CLASSES for v in out["classifications"].values())
            return utils.detect_raw(out["classifications"], out["regressions"], anchors, NUM_CLASSES, score_threshold=0.0105,
                                    capacity=int(rows * 0.5), return_raw=True, logits=True)

    res = {"workload": "BASELINE.json configs[4]: ResNeXt-50-FPN %dx%d bs=%d, forward + decode + class-wise NMS over all %d anchors per "
                       "image, random-init weights, synthetic N(0,1) images" % (size, size, batch, 196416)}
    try:
        for dtype in ("f16", "f32"):
            layers.set_inference_dtype(dtype)
            o = run(); torch.cuda.synchronize()
            counts = o[5].cpu().tolist()
            run(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                run()
            torch.cuda.synchronize()
            el_eager = (time.perf_counter() - t0) / iters
            # the same pass replayed from ONE hipGraph (static input buffer, static outputs: what a serving loop at a fixed shape
            # does; nothing on the host between the ~200 launches).  The eager figure is printed beside it; the graph's is the value
            # unless the capture fails.
            el, graphed = el_eager, False
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    og = run()
                g.replay(); torch.cuda.synchronize()
                if og[5].cpu().tolist()[:2] == counts[:2]:
                    t0 = time.perf_counter()
                    for _ in range(iters):
                        g.replay()
                    torch.cuda.synchronize()
                    el, graphed = (time.perf_counter() - t0) / iters, True
                del g, og
            except Exception:
                pass
            tf = INFERENCE_GFLOP_PER_IMAGE * batch / el / 1e3
            peak = FP16_MFMA_PEAK_TFLOPS if dtype == "f16" else FP32_MFMA_PEAK_TFLOPS
            res[dtype] = {"images_per_sec": round(batch / el, 2), "ms_per_batch": round(el * 1e3, 2), "hip_graph": graphed,
                          "images_per_sec_eager": round(batch / el_eager, 2), "candidates": counts[0],
                          "kept": counts[1], "conv_TFLOPs": round(tf, 1), "mfma_peak_TFLOPs": peak, "frac_of_mfma_peak": round(tf / peak, 4)}
    finally:
        layers.set_inference_dtype("f32")
    res["value"] = res["f16"]["images_per_sec"]
    res["unit"] = "images/sec (fp16)"
    return res


class _StandinAllReduce(object):
    """Single-GPU stand-in for the gradient all-reduce of an R-rank ring: where the trainer would issue
    the collective of an arena slice, a side stream runs `blocks` workgroups that stream 2 (R-1)/R x the slice's bytes at
    the ~150 GB/s one xGMI link gives a ring -- the CUs and the HBM traffic a collective takes from the backward pass that
    runs beside it.  Sums nothing (one rank)."""

    def __init__(self, arena, blocks, ranks=8, link_GBps=150.0):
        import _rn
        self._rn, self.arena, self.blocks, self.ranks, self.link = _rn, arena, blocks, ranks, link_GBps
        self.active, self.world, self.rank, self.launched = True, 1, 0, []
        self.capturable, self.host_staged = True, False                  # (a kernel on a private stream: recorded into the step's graph)
        self.stream = torch.cuda.Stream(device=arena.grads.device)       # (private: not one of _rn's joined side streams)
        self.sink = torch.empty_like(arena.grads)

    def launch(self, start=0, end=None):
        end = self.arena.count if end is None else end
        if end <= start:
            return
        moved = int(2 * (self.ranks - 1) / self.ranks * 4 * (end - start)) // 16 * 16
        moved = min(moved, 4 * (end - start) // 16 * 16)       # (one pass over the slice; the pacing sets the duration)
        us = 2 * (self.ranks - 1) / self.ranks * 4 * (end - start) / (self.link * 1e3)
        self.stream.wait_stream(torch.cuda.current_stream())
        self._rn.check(self._rn.lib().rn_debug_collective_standin(self.arena.grads[start:].data_ptr(), self.sink[start:].data_ptr(), moved,
                                                                  self.blocks, us, self.stream.cuda_stream), "rn_debug_collective_standin")
        self.launched.append((start, end))

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        return 1.0


def collective_standin(step, steps=60, warmup=10):
    """Step time with a ring all-reduce's footprint running under the backbone's backward pass, on ONE GPU: NOT a scaling
    measurement (no second rank, no xGMI) -- it bounds what sharing CUs / HBM with the collective kernels costs the step."""
    tr = step.trainer
    real = tr.allreduce

    def rate():
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    out = {"what": "8-rank ring stand-in: 2*(R-1)/R x slice bytes paced at 150 GB/s on a side stream, under segment B (heads + FPN slice) and "
                   "after it (backbone slice); single GPU, nothing is summed", "ms_per_step_without": round(rate(), 4)}
    try:
        for blocks in (16, 32, 64):
            tr.allreduce = _StandinAllReduce(tr.arena, blocks)
            ms = rate()
            out["blocks_%d" % blocks] = {"ms_per_step": round(ms, 4), "delta_ms": round(ms - out["ms_per_step_without"], 4)}
    finally:
        tr.allreduce = real
    return out


def collective_path_n1(args, device, plain_ips, steps=100, warmup=20):
    """The step several ranks run, on ONE rank: an RCCL process group of world size 1 with the collectives issued (nodes of the
    step's graph), the FPN / subnets split of the heads' slice behind the deferred weight-gradient products, MobileNetV2's
    stage cut.  What a scaling efficiency should be read against: the N = 1 `value` carries no collective and no stage cut."""
    import torch.distributed as dist
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        step = Step(device, use_graph=not args.no_graph, loss_mode=args.loss, dropout=args.dropout, rank=0, force_collective=True)
        step.trainer.check_interval = 0
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        tr = step.trainer
        ips = BATCH * steps / el
        out = {"images_per_sec": round(ips, 2), "ms_per_step": round(1e3 * el / steps, 4), "of_plain": round(ips / plain_ips, 4),
               "whole_step_in_one_graph": bool(tr._graphs and tr._graphs[5]), "collectives_captured_in_graph": bool(tr.allreduce.capturable),
               "tower_weight_gradients_deferred": bool(tr.defer_wgrad), "slice_schedule_bytes": [4 * (hi - lo) for lo, hi in tr.schedule]}
        del step, tr
    finally:
        if own:
            dist.destroy_process_group()
    torch.cuda.empty_cache()
    return out


def allreduce_slices(trainer, iters=20):
    """Per-slice RCCL all-reduce time alone (HIP events on the launch stream around `iters` back-to-back collectives) and
    its bus bandwidth 2 (R-1)/R x bytes / time: what one xGMI ring sustains for the messages this step sends."""
    import torch.distributed as dist
    ar = trainer.allreduce
    ranges = [tuple(r) for r in trainer.schedule] or [(0, trainer.arena.count)]
    buf = torch.zeros_like(trainer.arena.grads)
    out = []
    for lo, hi in ranges:
        if hi <= lo:
            continue
        for _ in range(3):
            dist.all_reduce(buf[lo:hi])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_reduce(buf[lo:hi])
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / iters
        nbytes = 4 * (hi - lo)
        out.append({"bytes": nbytes, "ms_alone": round(ms, 4),
                    "bus_GBps": round(2 * (ar.world - 1) / ar.world * nbytes / (ms * 1e6), 1)})
    return out


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


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
    ap.add_argument("--no-extras", action="store_true", help="skip the single-GPU extras (collective stand-in, fresh-data loop, cfg 1 / 3 / 4, cfg-5 inference)")
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

    import ops

    def run_once(capture_collectives=None, steps=None, warmup=None):
        """Build the step, warm up, time exactly args.steps steps; returns (step, elapsed, exposed, losses, gn_timeouts)."""
        steps = args.steps if steps is None else steps
        step = Step(device, use_graph=not args.no_graph, loss_mode=args.loss, dropout=args.dropout, rank=rank,
                    overlap=not args.no_overlap, force_collective=args.force_collective, capture_collectives=capture_collectives)
        step.trainer.check_interval = 0            # checked explicitly below (a timeout switches the path, it does not abort)
        for _ in range(args.warmup if warmup is None else warmup):
            step()

        def barrier():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        step.trainer.timing = {}
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        barrier()
        elapsed = time.perf_counter() - t0
        exposed = step.trainer.allreduce_exposed_ms()
        timeouts = _rn.barrier_timeouts()
        if dist is not None:
            t = torch.tensor([elapsed, exposed, float(timeouts)], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, exposed, timeouts = float(t[0].item()), float(t[1].item()), int(t[2].item())
        return step, elapsed, exposed, [float(x) for x in out], timeouts

    step, elapsed, exposed, losses, timeouts = run_once()
    gn_fallback = False
    if os.environ.get("RN_BENCH_FORCE_GN_FALLBACK") == "1":     # self-test of the fallback below
        timeouts = 1
    if timeouts:
        # a grid-resident GroupNorm block waited for a peer that was not co-resident (other kernels held CUs: a collective
        # under the backward pass, another process): its results are invalid.  Switch to the launch-ordered GroupNorm
        # kernels on EVERY rank and measure again -- the number reported is then the one of the path that is correct here.
        ops.GN_GRID_RESIDENT = False
        _rn.reset_barrier_timeouts()
        gn_fallback = True
        del step
        step, elapsed, exposed, losses, timeouts = run_once()
        if timeouts:
            raise SystemExit("bench.py: GroupNorm exchange timeouts with the grid-resident path off: invalid run")

    # multi-GPU evidence, collected on every rank (collectives) before rank 0 prints: per-rank exposed time, per-slice bus rate
    dist_info = None
    segmented = None
    if dist is not None and step.trainer.allreduce.active and step.trainer._graphs and step.trainer._graphs[5]:
        # the collectives are nodes of the step's graph: no event can be put around the final wait.  The time the compute stream
        # waits for collectives after the last backward kernel is measured on the OTHER path -- one graph per part, eager
        # collectives between them (what runs where RCCL's all-reduce cannot be captured) -- in a short second run, every rank
        # taking part; its rate is printed beside the headline as the A/B of the two paths on this node
        st2, el2, ex2, _l2, _t2 = run_once(capture_collectives=False, steps=min(args.steps, 40), warmup=min(args.warmup, 10))
        segmented = {"images_per_sec": round(world * BATCH * min(args.steps, 40) / el2, 2), "allreduce_exposed_ms": round(ex2, 4),
                     "slice_schedule_bytes": [4 * (hi - lo) for lo, hi in st2.trainer.schedule]}
        exposed_tr = st2.trainer
    else:
        exposed_tr = step.trainer
    if dist is not None:
        per_rank = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([exposed_tr.allreduce_exposed_ms()], dtype=torch.float64, device=device))
        per_rank = [float(t.item()) for t in per_rank]
        dist_info = {"rccl_ranks": world, "allreduce_exposed_ms_per_rank": [round(x, 4) for x in per_rank],
                     "allreduce_exposed_ms_max": round(max(per_rank), 4), "allreduce_exposed_ms_mean": round(sum(per_rank) / world, 4),
                     "slices": allreduce_slices(step.trainer)}
        if segmented is not None:
            dist_info["one_graph_per_part_eager_collectives"] = segmented
            exposed = segmented["allreduce_exposed_ms"]
        del exposed_tr

    result = None
    # the gradient slices in the order the trainer reduces them: heads + FPN after segment A, then one per part of the backbone's
    # backward pass (MobileNetV2 has two parts while a collective is active: cut behind the C3 tap)
    tr_ = step.trainer
    sched = [tuple(r) for r in tr_.schedule] or [(0, tr_.arena.count)]     # the slices in the order the last step issued them
    # one rank issues no collective and therefore takes no cut inside MobileNetV2's chain (it costs two kernels and hides
    # nothing): the schedule several ranks follow is printed beside this run's own
    multi = [4 * (hi - lo) for lo, hi in sched]
    bb_ = getattr(getattr(tr_, '_cut_base', None), 'backbone', None)
    if tr_.cut_offset and not tr_.allreduce.active and getattr(bb_, 'stage_cut_needs_collective', False):
        import mobilenet_v2
        first = next(iter(getattr(bb_, bb_.block_names[bb_.block_names.index(mobilenet_v2.STAGE_CUT_AFTER) + 1]).parameters()))
        off = tr_._param_offset[id(first)]
        # (FPN slice when segment A ends, the subnets' slice behind their deferred weight-gradient products, then the chain's two parts)
        multi = [4 * (tr_.heads_offset - tr_.cut_offset), 4 * (tr_.arena.count - tr_.heads_offset), 4 * (tr_.cut_offset - off), 4 * off]
    if rank == 0:
        ips = world * BATCH * args.steps / elapsed
        result = {
            "metric": "train images/sec (MobileNetV2-FPN RetinaNet 512x512, bs=2/GPU)",
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: MobileNetV2-FPN 512x512 bs=2/GPU, 80 classes, GroupNorm, "
                                   "%s + smooth-L1, dropout %.2f, momentum SGD, anchor assignment in the step" %
                                   (args.loss, args.dropout),
                       "global_batch": world * BATCH, "image_size": IMAGE_SIZE, "parallelism": "dp%d" % world,
                       "hip_graph": not args.no_graph,
                       "backward_segments": 1 + sum(1 for lo, hi in sched if hi <= tr_.cut_offset) if tr_.cut_offset else 1,
                       # segment A, the collectives (nodes of the graph when RCCL's all-reduce replays correctly from a captured
                       # graph: probed at start-up on every rank), every part of segment B and the update: ONE graph per step
                       "whole_step_in_one_graph": bool(tr_._graphs and len(tr_._graphs) > 5 and tr_._graphs[5]),
                       "collectives_captured_in_graph": bool(tr_.allreduce.active and tr_._graphs and tr_._graphs[5]),
                       "tower_weight_gradients_deferred": bool(tr_.defer_wgrad),
                       "arithmetic": ("fp32 storage, accumulation and results everywhere; the Winograd products (88 % of the multiply-adds) are evaluated "
                                      "on the bf16 matrix cores from EXACT three-way bf16 splits of the fp32 operands, six partial products per "
                                      "product: error <= ~2^-23 per elementary product, measured equal to the fp32 MFMA kernels' against fp64 "
                                      "(tests/test_gpu_x3.py); RN_PROD_X3=0 selects the fp32 matrix-core kernels") if _rn.lib().rn_get_product_mode()
                                     else "fp32 everywhere (exact fp32 matrix-core instruction)",
                       "parity": "this step at this dropout rate is oracle-checked with the kernels' counter-based masks injected at the "
                                 "reference's dropout sites (tests/test_gpu_fullsize.py::test_cfg2_full_size_train_step_matches_oracle[0.2], "
                                 "tests/test_gpu_dropout.py)",
                       "allreduce": {"backend": "rccl" if dist is not None else None, "ranks": world,
                                     "collectives_issued": bool(step.trainer.allreduce.active),
                                     # the slices in the order they are reduced; only the LAST one is reduced after the last backward kernel
                                     "slice_schedule_bytes": [4 * (hi - lo) for lo, hi in sched],
                                     "bytes_overlapped_with_backbone_backward": sum(4 * (hi - lo) for lo, hi in sched[:-1]),
                                     "bytes_after_backward": 4 * (sched[-1][1] - sched[-1][0]),
                                     # (with >= 2 ranks: tests/dist_worker.py part (c) runs exactly this schedule over RCCL)
                                     "slice_schedule_bytes_with_2_or_more_ranks": multi,
                                     "bytes_after_backward_with_2_or_more_ranks": multi[-1],
                                     "allreduce_exposed_ms": round(exposed, 4)},
                       "final_class_loss": round(losses[0], 6),
                       "final_regr_loss": round(losses[1], 6),
                       "mobilenet_chain": "rn_mb_* kernels (every GroupNorm applied by its consumer): all 17 bottlenecks + the output conv",
                       "gn_barrier_timeouts": timeouts, "gn_grid_resident": bool(ops.GN_GRID_RESIDENT),
                       "gn_fell_back_to_launch_ordered_kernels": gn_fallback,
                       # direct-convolution-equivalent FLOPs (what the reference's graph multiplies) vs the FLOPs the step EXECUTES
                       # after Winograd F(4x4,3x3) -- both against the dense fp32 MFMA peak
                       "conv_roofline_frac_whole_step": round(ips / world * TRAIN_GFLOP_PER_IMAGE / 1e3 /
                                                              FP32_MFMA_PEAK_TFLOPS, 4),
                       "executed_flop_frac_whole_step": round(ips / world * EXECUTED_TRAIN_GFLOP_PER_IMAGE / 1e3 /
                                                              FP32_MFMA_PEAK_TFLOPS, 4),
                       "executed_gflop_per_image": EXECUTED_TRAIN_GFLOP_PER_IMAGE,
                       "direct_equivalent_gflop_per_image": TRAIN_GFLOP_PER_IMAGE},
        }
        if dist_info is not None:
            result["config"]["allreduce"].update(dist_info)
        if not args.no_roofline:
            main, second, gn = roofline_kernels(device)
            result["roofline"] = dict(main, entries=[second, gn])
        if not args.no_nms:
            result["nms"] = nms_benchmark(device)
        if not args.no_extras and world == 1:
            # single-GPU extras (all driver-timed inside this run): the stand-in for a collective under the backward pass, the
            # product's own loop on fresh data, cfg 1 / 3 / 4 per-GPU workloads, cfg 5 inference
            if not args.no_graph:
                result["config"]["collective_standin"] = collective_standin(step)
            del step
            torch.cuda.empty_cache()
            if not args.no_graph and not started:
                try:
                    cp = collective_path_n1(args, device, ips)
                    result["config"]["allreduce"]["n1_collective_path"] = cp
                    result["config"]["allreduce"]["n1_collective_path_images_per_sec"] = cp["images_per_sec"]
                except Exception as e:      # noqa: BLE001 -- an extra must not take the headline line down (no RCCL group on this box ...)
                    result["config"]["allreduce"]["n1_collective_path"] = {"error": "%s: %s" % (type(e).__name__, e)}
                    result["config"]["allreduce"]["n1_collective_path_images_per_sec"] = None
            result["config"]["fresh_data"] = fresh_data(device, ips)
            result["config"]["cfg1_gpu"] = cfg1_gpu(device)
            torch.cuda.empty_cache()
            result["other_configs"] = {"note": "per-GPU workloads of BASELINE configs[2] and configs[3] on ONE GPU (their multi-GPU runs are the driver's)",
                                       "cfg3": other_config(device, "resnet_50", 800, 2), "cfg4": other_config(device, "densenet_121", 640, 4)}
            torch.cuda.empty_cache()
            result["inference"] = inference_benchmark(device)
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline()
    # RCCL prints a version banner through C stdio when a communicator is created; on a pipe that buffer would be flushed at
    # exit, AFTER the line below -- flush it now on every rank so that the JSON line is the last thing on stdout
    _flush_c_stdio()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        _flush_c_stdio()


if __name__ == "__main__":
    main()
