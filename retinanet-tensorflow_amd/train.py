"""Training step for the RetinaNet hot path (step semantics of reference train.py:111-134,
206-243, 261-273) -- the Estimator / CLI / summary plumbing around it is out of scope.

  model_fn composition   train.py:206-243  ->  Trainer.forward_backward(): net -> process_labels_and_logits
                                               -> losses.loss -> (+ L2 regulariser, folded into the
                                               optimizer kernel as wd*w, value from rn_grad_norm_l2reg)
  build_train_step       train.py:111-134  ->  Trainer.apply(): fused optimizer kernel on one flat arena
  MirroredStrategy       train.py:261-267  ->  one process per GPU; gradients averaged with ONE RCCL
                                               all-reduce per bucket over xGMI (torch.distributed 'nccl')

MI355X-first design: all parameters, gradients and optimizer slots live in three flat fp32
arenas (one allocation each, per-parameter padding to 1024 elements) so that the optimizer is a
single kernel, the gradient all-reduce is a few large contiguous messages sized for per-link
xGMI bandwidth, and the whole step (forward, loss, backward, optimizer) can be captured into one
hipGraph -- the launch-bound small kernels of the pyramid's coarse levels then cost no host time.
"""
import collections
import contextlib
import ctypes
import os
import sys

import numpy as np
import torch

import _rn
import layers as L
import losses
import ops
import utils
from levels import build_levels

OPT_BLOCK = _rn.OPT_BLOCK
FUSED_OPT_NORM = os.environ.get("RN_FUSED_OPT_NORM", "1") == "1"     # (tuning aids; both parity-neutral)


class ParamArena(object):
    """Flat fp32 storage for every parameter of `model` (and its gradients).

    Each parameter is re-pointed to a view of `self.weights`; `param.grad` is a view of
    `self.grads`.  Every parameter starts on a 1024-element boundary so the optimizer / norm
    kernels can look the L2 scale up per block."""

    def __init__(self, model, device):
        params = [p for p in model.parameters() if p.requires_grad]
        sizes = [p.numel() for p in params]
        padded = [(s + OPT_BLOCK - 1) // OPT_BLOCK * OPT_BLOCK for s in sizes]
        self.count = int(sum(padded))
        self.num_params = int(sum(sizes))
        self.weights = torch.zeros(self.count, dtype=torch.float32, device=device)
        self.grads = torch.zeros(self.count, dtype=torch.float32, device=device)
        wd = np.zeros(self.count // OPT_BLOCK, dtype=np.float32)
        self.offsets = []
        off = 0
        for p, s, ps in zip(params, sizes, padded):
            view = self.weights[off:off + s].view(p.shape)
            view.copy_(p.data.to(device))
            scale = float(getattr(p, 'l2_scale', 0.0))
            p.data = view
            p.grad = self.grads[off:off + s].view(p.shape)
            wd[off // OPT_BLOCK:(off + ps) // OPT_BLOCK] = scale
            self.offsets.append((off, s))
            off += ps
        self.params = params
        self.wd_per_block = torch.from_numpy(wd).to(device)

    def zero_grad(self):
        if self.grads.is_cuda:          # our own kernel: no PyTorch kernel inside the step
            _rn.check(_rn.lib().rn_zero(_rn.f32(self.grads), self.grads.numel(), _rn.stream()), "rn_zero")
        else:
            self.grads.zero_()


class Optimizer(object):
    """tf.train.MomentumOptimizer(lr, 0.9) | RMSPropOptimizer(lr, 0.9, 0.9) | AdamOptimizer(lr)
    (train.py:114-119) + optional tf.clip_by_global_norm (train.py:127-132), one fused kernel."""

    def __init__(self, arena, kind='momentum', learning_rate=1e-2, grad_clip_norm=None):
        assert kind in ['momentum', 'adam', 'rmsprop']
        self.arena, self.kind, self.lr = arena, kind, float(learning_rate)
        self.clip = float(grad_clip_norm) if grad_clip_norm is not None else 0.0
        dev = arena.weights.device
        self.state1 = torch.ones_like(arena.weights) if kind == 'rmsprop' else torch.zeros_like(arena.weights)
        self.state2 = torch.zeros_like(arena.weights) if kind != 'momentum' else None
        self.norm_reg = torch.zeros(2, dtype=torch.float32, device=dev)   # [sum g'^2, L2 reg loss]
        # (sum g'^2, regulariser) pairs of the slices of a step: allocated HERE, not lazily in begin_step -- a first-step zero-fill
        # on the main stream would not be ordered against a side stream's slice update that writes its pairs
        self._pairs = 0
        self._partial = None
        if dev.type == 'cuda':
            n = 4 * int(_rn.lib().rn_optimizer_norm_pairs(arena.count)) + 16
            self._partial = torch.zeros(2 * n, dtype=torch.float64, device=dev)
        self.step_count = 0

    def step(self, grad_scale=1.0, advance_counter=None):
        """`advance_counter`: the device word the dropout masks hash; bumped by the optimizer kernel itself."""
        a = self.arena
        if self.clip <= 0.0 and a.weights.is_cuda and FUSED_OPT_NORM:
            # no clipping: the norm is not an input of the update -- one pass over the arena forms it beside the update
            self.begin_step()
            self.step_slice(0, a.count, grad_scale, advance_counter)
            self.finish_step()
            return
        L_ = _rn.lib()
        ws = _rn.workspace(L_.rn_optimizer_workspace(a.count), a.weights.device)
        _rn.check(L_.rn_grad_norm_l2reg(_rn.f32(a.weights), _rn.f32(a.grads), _rn.f32(a.wd_per_block), a.count,
                                        grad_scale, _rn.f32(self.norm_reg), ws.data_ptr(), ws.numel(), _rn.stream()),
                  'rn_grad_norm_l2reg')
        self.step_count += 1
        _rn.check(L_.rn_optimizer_step(_rn.OPT[self.kind], _rn.f32(a.weights), _rn.f32(a.grads), _rn.f32(self.state1),
                                       _rn.f32(self.state2) if self.state2 is not None else None,
                                       _rn.f32(a.wd_per_block), a.count, self.lr, grad_scale, self.clip,
                                       _rn.f32(self.norm_reg), self.step_count,
                                       advance_counter.data_ptr() if advance_counter is not None else None,
                                       ops.DROPOUT_COUNTER_STEP, _rn.stream()), 'rn_optimizer_step')
        import ops_f16
        ops_f16.weights_changed()      # fp16-packed copies of the kernels (inference path) are stale now

    # -- the update in slices of the arena (no clipping): Trainer.step updates the heads + FPN slice on a side stream while the
    # backbone's backward pass still runs, the rest after it; every slice's launch leaves its share of (sum g'^2, regulariser)
    def begin_step(self):
        self.step_count += 1
        self._pairs = 0
        assert self._partial is not None, "the fused norm path needs a device arena"

    def step_slice(self, lo, hi, grad_scale, advance_counter=None, stream=None):
        a, L_ = self.arena, _rn.lib()
        assert 0 <= lo < hi <= a.count and lo % OPT_BLOCK == 0 and hi % OPT_BLOCK == 0
        npairs = int(L_.rn_optimizer_norm_pairs(hi - lo))
        assert 2 * (self._pairs + npairs) <= self._partial.numel()
        part = self._partial[2 * self._pairs:]
        self._pairs += npairs
        _rn.check(L_.rn_optimizer_step_norm(_rn.OPT[self.kind], a.weights[lo:].data_ptr(), a.grads[lo:].data_ptr(), self.state1[lo:].data_ptr(),
                                            self.state2[lo:].data_ptr() if self.state2 is not None else None,
                                            a.wd_per_block[lo // OPT_BLOCK:].data_ptr(), hi - lo, self.lr, grad_scale, self.step_count,
                                            advance_counter.data_ptr() if advance_counter is not None else None, ops.DROPOUT_COUNTER_STEP,
                                            part.data_ptr(), stream if stream is not None else _rn.stream()), 'rn_optimizer_step_norm')

    def finish_step(self):
        _rn.check(_rn.lib().rn_norm_reg_finalize(self._partial.data_ptr(), self._pairs, _rn.f32(self.norm_reg), _rn.stream()),
                  'rn_norm_reg_finalize')
        import ops_f16
        ops_f16.weights_changed()

    @property
    def regularization_loss(self):
        """sum_w scale * ||w||^2 / 2 at the weights used for the last gradient (train.py:221)."""
        return self.norm_reg[1]


class GradientAllReduce(object):
    """MirroredStrategy's cross-replica gradient sum (train.py:261-267) as RCCL all-reduces of
    contiguous arena slices.  xGMI is point-to-point (7 links x ~153 GB/s), ring collectives are
    per-link bound, so a region goes out as FEW LARGE buckets (default 64 MB: the whole 40 MB arena of the
    headline config would be ONE message) rather than one message per tensor; the 1/world_size average is
    folded into the optimizer kernel.

    launch(start, end) enqueues the (asynchronous) all-reduce of grads[start:end] behind the work already
    queued on the current stream and returns at once: the collective runs on the process group's own
    stream while the current stream goes on with the next backward segment; wait() joins."""

    def __init__(self, arena, process_group=None, bucket_bytes=64 << 20, force=False):
        import torch.distributed as dist
        self.dist, self.group, self.arena = dist, process_group, arena
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # force: issue the collectives even with one rank (self-tests of the RCCL path on a 1-GPU box)
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.per = max(OPT_BLOCK, (int(bucket_bytes) // 4) // OPT_BLOCK * OPT_BLOCK)
        # a backend without device collectives (gloo: debugging, tests with several processes on one GPU) gets the
        # bucket through host memory, synchronously; the production backend ('nccl' = RCCL) reduces in place in HBM
        self.host_staged = bool(self.active and arena.grads.is_cuda and dist.get_backend(process_group) != 'nccl')
        self.buckets = self.buckets_of(0, arena.count)
        self._works = []
        self.launched = []          # (start, end) of every bucket issued since the last wait(): for tests
        self._capturable = None     # decided once, by doing it: probe_capturable()

    def buckets_of(self, start, end):
        return [(s, min(s + self.per, end)) for s in range(start, end, self.per)]

    @property
    def capturable(self):
        """May launch() / wait() be recorded into a hipGraph (the collectives as nodes of the step's graph)?"""
        return bool(self._capturable)

    def probe_capturable(self, device):
        """Decide -- by doing it, on every rank, with a known answer -- whether this backend's all-reduce can be CAPTURED into a
        hipGraph and replayed: rank r contributes r + 1, a replay must leave world (world + 1) / 2 everywhere, twice (a second
        replay must not reduce the first one's result again).  An exception or a wrong sum on ANY rank (MIN over ranks, eager
        collective) selects the eager collectives between captured segments on EVERY rank.  RN_DP_CAPTURE=0 skips the probe
        (never capture), =1 / auto (default) runs it."""
        if self._capturable is not None:
            return self._capturable
        self._capturable = False
        if (not self.active or self.host_staged or device.type != 'cuda' or os.environ.get("RN_DP_CAPTURE", "auto") == "0"):
            return False
        ok, why = 1.0, ""
        try:
            cur = torch.cuda.current_stream(device)
            buf = torch.full((OPT_BLOCK,), float(self.rank + 1), dtype=torch.float32, device=device)
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)     # eager first: the communicator connects outside any capture
            torch.cuda.synchronize(device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                w = self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                w.wait()
            want = float(self.world * (self.world + 1) // 2)
            for _ in range(2):
                buf.fill_(float(self.rank + 1))
                g.replay()
                cur.synchronize()
                if not bool((buf == want).all().item()):
                    ok, why = 0.0, "replayed all-reduce gave %r, expected %r" % (float(buf[0].item()), want)
            del g
        except Exception as e:       # noqa: BLE001 -- any failure means "do not capture", never "abort the run"
            ok, why = 0.0, "%s: %s" % (type(e).__name__, e)
        t = torch.tensor([ok], dtype=torch.float32, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        self._capturable = bool(t.item() == 1.0)
        if not self._capturable and self.rank == 0:
            print("[trainer] collectives are not captured into the step's graph (%s): eager collectives between captured segments"
                  % (why or "another rank's probe failed"), file=sys.stderr, flush=True)
        return self._capturable

    def launch(self, start=0, end=None):
        end = self.arena.count if end is None else end
        if not self.active or end <= start:
            return
        # last bucket first: within a region the later layers' gradients were produced first
        for s, e in reversed(self.buckets_of(start, end)):
            if self.host_staged:
                host = self.arena.grads[s:e].cpu()
                self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
                self.arena.grads[s:e].copy_(host)
                self.launched.append((s, e))
                continue
            self._works.append(self.dist.all_reduce(self.arena.grads[s:e], op=self.dist.ReduceOp.SUM, group=self.group,
                                                    async_op=True))
            self.launched.append((s, e))

    def wait(self):
        for w in self._works:
            w.wait()
        del self._works[:]
        return 1.0 / self.world

    def wait_on_current_stream(self):
        """Make the CURRENT stream wait for every collective issued so far (Work.wait() orders the stream, not the host) without
        forgetting them: wait() on the main stream later still covers them."""
        for w in self._works:
            w.wait()

    def __call__(self):
        self.launch(0, self.arena.count)
        return self.wait()


class Trainer(object):
    """One data-parallel replica.  `step(features)` = forward + loss + backward + gradient
    all-reduce + optimizer.

    Backward runs in TWO segments (`overlap=True`): (A) forward, loss and the backward pass of the heads + FPN --
    82 % of the gradient bytes of the headline config, complete first -- then (B) the backward pass of the
    backbone.  The all-reduce of region A's slice of the gradient arena is launched between the two and runs on
    RCCL's stream underneath segment B; only the (small) backbone slice is reduced after it.  The cut is made
    with detached leaves at the backbone taps the FPN reads (RetinaNetBase.backward_cut), so the arithmetic is
    unchanged.  With use_graph=True each segment is one hipGraph (shared memory pool, always replayed A then
    B); the collectives stay eager launches between the replays."""

    def __init__(self, net, levels=None, optimizer='momentum', learning_rate=1e-2, grad_clip_norm=None,
                 loss_mode='bce_dice', device='cuda', use_graph=False, process_group=None,
                 direct_param_grads=True, wgrad_side_stream=False, defer_reductions=True, overlap=True,
                 force_collective=False, input_fn=None, check_interval=50, capture_collectives=None):
        self.net, self.levels = net, levels or build_levels()
        self.device = torch.device(device)
        if self.device.type == 'cuda':
            if self.device.index is None:
                self.device = torch.device('cuda', torch.cuda.current_device())
            # the kernels are launched on the CURRENT device's current stream with raw pointers
            assert torch.cuda.current_device() == self.device.index, \
                "call torch.cuda.set_device(%d) before building the Trainer" % self.device.index
        # (the optimizer kernel is launched per step, outside the captured segments, with the host's step count: Adam's
        # bias-corrected learning rate is a fresh launch argument every step, also with use_graph=True)
        self.loss_mode = loss_mode
        self.arena = ParamArena(net, self.device)
        self.opt = Optimizer(self.arena, optimizer, learning_rate, grad_clip_norm)
        self.allreduce = GradientAllReduce(self.arena, process_group, force=force_collective)
        self.use_graph = use_graph
        self.input_fn = input_fn       # optional: features = input_fn(), run INSIDE segment A (e.g. device-side label assignment)
        self.defer_reductions = (bool(defer_reductions) and bool(direct_param_grads) and
                                 os.environ.get("RN_DEFER_REDUCTIONS", "1") == "1")
        # kernels write parameter gradients straight into the arena (every parameter of this network is used by exactly one
        # op call per step; ops._grad_slot refuses a second write); see ops.DIRECT_PARAM_GRADS.  These switches and the dropout
        # counter are process-wide in ops / layers: they are set only while THIS trainer's segments run (_scoped) and restored
        # afterwards, so trainers and plain autograd users in one process do not see each other's settings
        self.direct_param_grads = bool(direct_param_grads)
        # ... which would let the weight-gradient kernels run on a side stream under the dgrad / GroupNorm chain;
        # measured on MI355X: 162 vs 170 img/s (fork/join edges + CU contention cost more than the overlap buys), so off
        self.wgrad_side_stream = bool(direct_param_grads) and bool(wgrad_side_stream)
        # backward segments: arena offset where the FPN's parameters start (arena order = registration order =
        # backbone, fpn, classification_subnet, regression_subnet)
        self.cut_offset = 0
        base = getattr(net, 'base', None)
        if overlap and base is not None and hasattr(base, 'backward_cut') and hasattr(base, 'fpn'):
            first = next(iter(base.fpn.parameters()), None)
            for p, (off, _) in zip(self.arena.params, self.arena.offsets):
                if p is first:
                    self.cut_offset = off
        # (the hook itself -- base.backward_cut -- is installed only while one of this trainer's segments runs: _scoped)
        self._cut_base = base if self.cut_offset else None
        self._cut_src = self._cut_leaves = None
        # ... and the backbone's own backward pass in one part per stage where the backbone offers the cut points (`stage_cut`:
        # ResNeXt / DenseNet, whose 200 / 50 MB of backbone gradients would otherwise all wait for the last backward kernel):
        # part j's slice of the gradient arena is reduced underneath the parts that follow it
        bb = getattr(base, 'backbone', None) if self._cut_base is not None else None
        self._stage_bb = bb if (bb is not None and hasattr(bb, 'stage_cut') and os.environ.get("RN_STAGE_CUTS", "1") == "1") else None
        # (a backbone whose cut costs kernels -- MobileNetV2's chain -- takes it only where a collective is there to hide)
        if self._stage_bb is not None and getattr(bb, 'stage_cut_needs_collective', False) and not self.allreduce.active:
            self._stage_bb = None
        self._stage_cuts = []          # per step: (arena offset, source tensor, detached leaf, names of the taps made before it)
        self._param_offset = {id(p): off for p, (off, _) in zip(self.arena.params, self.arena.offsets)}
        self._parts = []               # per step: (roots, grads-of-leaves getter, arena range) of segment B's parts
        # The head towers' WEIGHT gradients (half of their backward products) beside the backbone's backward pass instead of inside
        # the heads' (ops.WGRAD_DEFER): segment A records them, segment B launches them on a side stream before its own kernels.
        # With several ranks the step keeps them there (round 6): the slice that is complete when segment A ends -- the FPN's,
        # [cut_offset, heads_offset) -- is all-reduced at once, the subnets' slice [heads_offset, count) BEHIND the deferred products
        # on their stream (the collective waits for that stream, not for the backbone's backward pass).  A forked stream must be
        # joined inside the graph that forked it, so with one graph PER PART and eager collectives between them (the fallback when
        # the collectives cannot be captured) the products are forked AND joined inside the first part's graph and the subnets'
        # slice is reduced right behind that part, under the parts that follow.  RN_DEFER_WGRAD=0: off everywhere.
        self.heads_offset = self.arena.count
        if base is not None and hasattr(base, 'classification_subnet'):
            first = next(iter(base.classification_subnet.parameters()), None)
            self.heads_offset = self._param_offset.get(id(first), self.arena.count)
        # One captured graph per step needs nothing between its parts -- or collectives that can themselves be recorded into the
        # graph (GradientAllReduce.probe_capturable decides that by doing it, identically on every rank).
        self.whole_step_graph = os.environ.get("RN_WHOLE_STEP_GRAPH", "1") == "1"
        if capture_collectives is False:          # (A/B aid and fallback: eager collectives between one graph per part)
            self.allreduce._capturable = False
        if self.allreduce.active and use_graph and self.whole_step_graph and self.device.type == 'cuda':
            self.allreduce.probe_capturable(self.device)
        self.defer_wgrad = os.environ.get("RN_DEFER_WGRAD", "1") == "1" and self.device.type == 'cuda' and self.direct_param_grads
        # OPT-IN (RN_WINO_PRE=1; measured SLOWER, kept as the record of the experiment): the Winograd kernel transforms of the head
        # towers' 10 convs once per step on a side stream beside the backbone's forward pass, instead of inside every layer's first
        # launch (ops.WinoPretransform).  Same kernels, same values -- the weights do not change between the optimizer's update and
        # the forward pass that follows -- and the layers' launches lose their kernel-transform blocks (upper bound with the
        # transforms skipped altogether: 491 -> 497 - 502 images/s), but ten more launches beside the backbone's latency-bound
        # chain cost more than that: 471 - 485 against 488 - 494 images/s (profiles/r06_ab_runs.txt).
        self.wino_pre = os.environ.get("RN_WINO_PRE", "0") == "1" and self.device.type == 'cuda'
        self._wino_pre = None
        self._deferred_wgrads = []
        self._deferred_running = []
        self._graphs = None
        # input shape key (dataset.DeviceFeed.shape_key; None without a feed) -> captured segments, least recently used first.
        # Every entry owns its graphs' private activation pool and static buffers, so the cache is BOUNDED (RN_GRAPH_CACHE, default 4
        # shapes): a loader with many raw sizes should bucket / pad them; a key beyond the bound evicts the least recently used set
        # (its next appearance re-captures: two warm-up passes + capture, logged).
        self._graph_cache = collections.OrderedDict()
        self.graph_cache_max = max(1, int(os.environ.get("RN_GRAPH_CACHE", "4")))
        # The whole step -- segment A, the collectives, every part of segment B, the update -- is ONE captured graph wherever
        # nothing has to happen on the host between its parts: the ~80 us a step the GPU idles at the graph boundaries and in
        # front of an eagerly launched update (profiles/r05_bench_step_timeline.txt: "idle stretches") disappear.  The update's
        # host-side scalars are constants of such a graph: momentum SGD without clipping only (no step-dependent scalar), and a
        # changed learning rate captures again (the rate is part of the cache key).  RN_WHOLE_STEP_GRAPH=0: one graph per part.
        self.schedule = []             # the gradient slices of the last recorded / eager step, in the order their all-reduce was issued
        self._last_lr, self._lr_changes = None, 0
        self.recaptures = 0
        self._static = None
        # the device word every fused dropout hashes (bumped by the optimizer kernel); replicas start it at different values
        # so that they draw different masks (each tower of the reference's MirroredStrategy has its own dropout stream)
        self.drop_rank_offset = self.allreduce.rank * 0x632BE59BD9B4E019 % (1 << 62)      # (checkpoints store counter - offset)
        self.drop_counter = torch.full((1,), self.drop_rank_offset, dtype=torch.int64, device=self.device)
        self._one = torch.ones((), dtype=torch.float32, device=self.device)
        self.check_interval = int(check_interval)
        self.steps_done = 0
        self.timing = None             # set to {} to collect 'allreduce_exposed_ms' (events around the final wait)
        self.last = {}

    @contextlib.contextmanager
    def _scoped(self):
        """This trainer's process-wide switches (direct parameter gradients, weight-gradient side stream, dropout counter) for
        the duration of one of its segments."""
        saved = (ops.DIRECT_PARAM_GRADS, ops.WGRAD_SIDE_STREAM, L.Dropout.seed_device_counter)
        ops.DIRECT_PARAM_GRADS, ops.WGRAD_SIDE_STREAM = self.direct_param_grads, self.wgrad_side_stream
        L.Dropout.seed_device_counter = self.drop_counter
        base = self._cut_base
        saved_cut = base.backward_cut if base is not None else None
        saved_stage = self._stage_bb.stage_cut if self._stage_bb is not None else None
        if base is not None:        # plain autograd users of the same net (and other trainers) never see this trainer's cut
            base.backward_cut = self._cut
        if self._stage_bb is not None:
            self._stage_bb.stage_cut = self._stage_cut
        try:
            yield
        finally:
            ops.DIRECT_PARAM_GRADS, ops.WGRAD_SIDE_STREAM, L.Dropout.seed_device_counter = saved
            if base is not None:
                base.backward_cut = saved_cut
            if self._stage_bb is not None:
                self._stage_bb.stage_cut = saved_stage

    # -- replica-local work (capturable)
    def _cut(self, taps):
        """RetinaNetBase.backward_cut: the FPN reads detached leaves; segment B feeds their gradients back."""
        keys = [k for k in ('C3', 'C4', 'C5') if k in taps]
        self._cut_keys = keys
        self._cut_src = [taps[k] for k in keys]
        self._cut_leaves = [t.detach().requires_grad_(True) for t in self._cut_src]
        return {**taps, **dict(zip(keys, self._cut_leaves))}

    def _stage_cut(self, module, x, taps_before=()):
        """<backbone>.stage_cut: the stage that starts with `module` reads a detached leaf (see __init__)."""
        first = next(iter(module.parameters()), None)
        off = self._param_offset.get(id(first)) if first is not None else None
        if off is None or not x.requires_grad or not (0 < off < self.cut_offset):
            return x
        leaf = x.detach().requires_grad_(True)
        self._stage_cuts.append((off, x, leaf, frozenset(taps_before)))
        return leaf

    def _plan_parts(self):
        """Segment B as parts, last stage first: [(roots, leaves whose .grad seeds them, (start, end) of the arena slice complete after it)]."""
        keys, srcs, leaves = self._cut_keys, self._cut_src, self._cut_leaves
        cuts = sorted(self._stage_cuts, key=lambda c: c[0])
        parts, done, hi = [], set(), self.cut_offset
        nxt = None                                     # (source, leaf) of the cut above the part being built
        for off, x, leaf, before in reversed(cuts):
            idx = [i for i, k in enumerate(keys) if k not in before and k not in done]
            done.update(keys[i] for i in idx)
            roots = [srcs[i] for i in idx] + ([nxt[0]] if nxt is not None else [])
            seeds = [leaves[i] for i in idx] + ([nxt[1]] if nxt is not None else [])
            parts.append((roots, seeds, (off, hi)))
            nxt, hi = (x, leaf), off
        idx = [i for i, k in enumerate(keys) if k not in done]
        roots = [srcs[i] for i in idx] + ([nxt[0]] if nxt is not None else [])
        seeds = [leaves[i] for i in idx] + ([nxt[1]] if nxt is not None else [])
        parts.append((roots, seeds, (0, hi)))
        return parts

    def _backward(self, roots, grads, skip_join=()):
        defer = self.defer_reductions and self.device.type == 'cuda' and ops.DIRECT_PARAM_GRADS
        if defer:                                  # ~130 gradient row reductions -> one launch per segment
            import retinanet
            extra = [retinanet.side_stream(self.device)] if retinanet.HEADS_TWO_STREAMS else []
            if ops.WGRAD_SIDE_STREAM:
                extra.append(_rn.side_stream(self.device, 0))
            if os.environ.get("RN_DEFER_SIDE", "1") != "1":
                extra = []
            ops.begin_deferred_reductions(extra)
        try:
            torch.autograd.backward(roots, grads)
        finally:
            if defer:
                ops.end_deferred_reductions()
        # weight-gradient kernels may run on side streams and write straight into the arena: join them before
        # anything (the collective, the optimizer) reads it
        if self.device.type == 'cuda':
            _rn.join_side_streams(self.device, skip=skip_join)

    def segment_a(self, features=None):
        """forward + loss + backward of the heads and the FPN (the whole backward pass when there is no cut)."""
        with self._scoped():
            label_stream = None
            zeroed = False
            if features is None:
                # input_fn builds the step's labels on the device (anchor assignment); only the loss reads them: their
                # kernels run on a side stream underneath the backbone's forward pass (the image itself -- written before
                # the step, not by input_fn -- is read at once).  OPT-IN (input_fn.concurrent = True): an input_fn that also
                # creates / uploads / preprocesses the IMAGE must stay on the main stream, or the backbone would read the image
                # with no ordering against the side stream that writes it.
                if self.device.type == 'cuda' and getattr(self.input_fn, 'concurrent', False):
                    label_stream = _rn.side_stream(self.device, 2)
                    label_stream.wait_stream(torch.cuda.current_stream())
                    main_stream = torch.cuda.current_stream()
                    with torch.cuda.stream(label_stream):
                        features = self.input_fn()
                        # (the gradient arena is cleared here too, underneath the backbone's forward pass, instead of between the
                        # loss and its gradient: nothing reads or writes it before the join in front of the loss)
                        self.arena.zero_grad()
                        zeroed = True
                    _for_each_tensor(features, lambda t: t.record_stream(main_stream) if t.is_cuda else None)   # (allocated on the side stream, read on this one)
                else:
                    features = self.input_fn()
            self._cut_src = self._cut_leaves = None
            self._stage_cuts = []
            self._parts = []
            ops.begin_direct_grad_step()       # a parameter's gradient slot may be written once per step from here on
            pre_ev = self._fork_kernel_transforms()
            try:
                logits = {'detection': self.net(features['image'], training=True)}
            finally:
                ops.WINO_PRE = {}
            if pre_ev is not None:             # (joined even if no layer read them: a captured side stream must rejoin)
                torch.cuda.current_stream().wait_event(pre_ev)
            if label_stream is not None:
                torch.cuda.current_stream().wait_stream(label_stream)
            inp, logits = utils.process_labels_and_logits(labels=features, logits=logits, levels=self.levels)
            class_loss, regr_loss = losses.loss(labels=inp['detection_trainable'], logits=logits['detection_trainable'],
                                                mode=self.loss_mode)
            # kernels that write a parameter's gradient directly overwrite it; gradients that reach a parameter
            # through autograd (e.g. the concatenated head kernels) are accumulated -> the arena starts at zero
            if not zeroed:
                self.arena.zero_grad()
            # d(class_loss + regr_loss): both roots seeded with the same pre-allocated 1 (no add / fill kernels in the step)
            ops.WGRAD_DEFER = [] if self.defer_wgrad else None
            try:
                self._backward([class_loss, regr_loss], [self._one, self._one])
            finally:
                self._deferred_wgrads, ops.WGRAD_DEFER = (ops.WGRAD_DEFER or []), None
            if self._cut_src is not None:
                self._parts = self._plan_parts()
                self._cut_src = self._cut_leaves = None
                self._stage_cuts = []
            if self._deferred_wgrads and not self._parts:       # no segment B to hide them under: finish them here
                ops.run_deferred_wgrads(self._deferred_wgrads)
                self._deferred_wgrads = []
            return class_loss.detach(), regr_loss.detach()

    PRE_STREAM = 8                     # _rn.side_stream index of the tower kernels' transforms

    def _fork_kernel_transforms(self):
        """Launch the head towers' kernel transforms on their side stream (behind everything queued so far: the optimizer's update of
        the previous step) and publish the buffers for this forward pass (ops.WINO_PRE).  Returns the event behind them, or None."""
        if not self.wino_pre or not (ops.WINO_GN_FOLD and ops.WINOGRAD):
            return None
        if self._wino_pre is None:
            if torch.cuda.is_current_stream_capturing():
                return None                    # (built by the eager steps in front of a capture; never allocate inside one)
            base = getattr(self.net, 'base', self.net)
            subnets = [getattr(base, n, None) for n in ('classification_subnet', 'regression_subnet')]
            kernels = [w for sn in subnets if sn is not None and hasattr(sn, 'tower_kernels') for w in sn.tower_kernels()]
            if not kernels:
                return None
            self._wino_pre = ops.WinoPretransform(kernels)
        side = _rn.side_stream(self.device, self.PRE_STREAM)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.WINO_PRE, ev = self._wino_pre.launch()
        return ev

    def num_parts(self):
        """Parts of segment B planned by the last segment_a (1 without stage cuts, 0 without any cut)."""
        return len(self._parts)

    WGRAD_STREAM = 7                   # _rn.side_stream index of the deferred weight-gradient products

    def segment_b(self, part=None, join_wgrads=True, after_wgrads=None):
        """backward of the backbone from the gradients segment A left at the cut: every part (part=None), or part j of
        num_parts() (last stage first).  Returns the (start, end) arena slice that is complete after it.

        The first call of a step forks the deferred tower weight gradients (segment A's records) onto their side stream;
        `after_wgrads()` runs on that stream behind them (the all-reduce of their slice); `join_wgrads=False` leaves the stream
        un-joined when the call returns -- the caller joins it (_join_wgrads) before anything reads those gradients."""
        if not self._parts:
            return None
        rng = None
        probe = int(os.environ.get("RN_PROBE_SIDE_PRODUCTS", "0"))   # tuning aid: N head-tower-sized batched products on a side
        side = None                                                   # stream beside the backbone's backward pass
        if probe and self.device.type == 'cuda':
            pm = int(os.environ.get("RN_PROBE_M", "682"))
            if not hasattr(self, "_probe_buf"):
                self._probe_buf = (torch.randn(36, pm, 256, device=self.device), torch.randn(36, 256, 256, device=self.device) * 0.01,
                                   torch.empty(36, pm, 256, device=self.device))
            A_, B_, C_ = self._probe_buf
            main, side = torch.cuda.current_stream(), _rn.side_stream(self.device, 6)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(probe):
                    _rn.check(_rn.lib().rn_gemm_batched(_rn.f32(A_), _rn.f32(B_), _rn.f32(C_), A_.shape[1], 256, 256, 36, 0, _rn.stream()), "rn_gemm_batched")
        if self._deferred_wgrads:             # the towers' weight-gradient halves: forked off here
            self._fork_wgrads(after_wgrads)
        skip = () if join_wgrads else (self.WGRAD_STREAM,)
        for j in (range(len(self._parts)) if part is None else [part]):
            roots, seeds, rng = self._parts[j]
            with self._scoped():
                self._backward(roots, [l.grad for l in seeds], skip_join=skip)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        if part is None:
            self._parts = []
            return (0, self.cut_offset)
        return rng

    def _fork_wgrads(self, after=None):
        if self.device.type == 'cuda':
            wside = _rn.side_stream(self.device, self.WGRAD_STREAM)
            wside.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(wside):
                ops.run_deferred_wgrads(self._deferred_wgrads)
                if after is not None:
                    after()
        else:                                 # (no streams: tests of the schedule on CPU)
            ops.run_deferred_wgrads(self._deferred_wgrads)
            if after is not None:
                after()
        self._deferred_running = self._deferred_wgrads      # (their workspaces stay allocated until the next step replaces this)
        self._deferred_wgrads = []

    def _join_wgrads(self):
        if self.device.type == 'cuda':
            torch.cuda.current_stream().wait_stream(_rn.side_stream(self.device, self.WGRAD_STREAM))

    def forward_backward(self, features=None, advance_dropout=True):
        """Both segments, no collective, no update.  The dropout counter is bumped here (one tiny launch); step() leaves
        that to the optimizer kernel."""
        out = self.segment_a(features)
        self.segment_b()
        if advance_dropout and self.device.type == 'cuda':
            ops.advance_dropout_counter(self.drop_counter)
        return out

    def _run_step(self, features=None, timed=False):
        """The device work of ONE step in issue order -- the same sequence whether it is launched eagerly or recorded into one
        graph, with one rank or many (MirroredStrategy's per-tower step + cross-tower sum, train.py:111-134, 261-267):

            segment A (assignment, forward, loss, backward of the heads and the FPN; tower weight gradients recorded, not run)
            all-reduce  [cut_offset, heads_offset)      the FPN's slice             } under segment B
            fork: deferred tower weight gradients  ->  all-reduce [heads_offset, count) behind them, on their stream
            for every part j of the backbone's backward pass, last stage first:  part j  ->  all-reduce of its slice
            join, wait for the collectives, optimizer update (1 / world folded in; bumps the dropout counter)

        Without deferred products the first collective covers [cut_offset, count).  With one rank launch() / wait() do nothing."""
        ar = self.allreduce
        del self.schedule[:]

        def reduce(lo, hi):
            if hi > lo:
                self.schedule.append((lo, hi))
                ar.launch(lo, hi)

        class_loss, regr_loss = self.segment_a(features)
        deferred = bool(self._deferred_wgrads) and bool(self._parts)
        top = self.heads_offset if (deferred and ar.active) else self.arena.count
        reduce(self.cut_offset, top)                       # heads + FPN (or the FPN alone): under the backbone's backward pass
        after = (lambda: reduce(top, self.arena.count)) if top < self.arena.count else None
        n = self.num_parts()
        for j in range(n if n else (1 if self.cut_offset else 0)):
            rng = self.segment_b(j, join_wgrads=False, after_wgrads=after) if n else None
            reduce(*(rng or (0, self.cut_offset)))         # this part's slice: under the parts that follow
        self._parts = []
        if deferred:
            self._join_wgrads()
        if timed and self.timing is not None and self.device.type == 'cuda':
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            grad_scale = ar.wait()
            e1.record()
            self.timing.setdefault('exposed_events', []).append((e0, e1))
        else:
            grad_scale = ar.wait()
        self.opt.step(grad_scale, self.drop_counter)       # also bumps the dropout counter: fresh masks next step
        return class_loss, regr_loss

    def _warm_up(self):
        # warm-up on a side stream (allocator + workspace sizing), then capture; ONE warm-up stream per trainer: every shape
        # key re-captures, and per-stream state elsewhere (_rn.sync_counters) is keyed by the stream handle
        s = getattr(self, "_warm_stream", None)
        if s is None:
            s = self._warm_stream = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            counter = self.drop_counter.clone()
            for _ in range(2):
                self.forward_backward(self._static, advance_dropout=False)
            self.drop_counter.copy_(counter)       # the warm-up passes do not count: replay i draws the masks eager step i draws
        torch.cuda.current_stream().wait_stream(s)

    def _capture(self, features):
        """One graph for segment A and one per part of segment B; collectives and the update are eager launches between / after
        the replays (several ranks whose collectives cannot be captured; RN_WHOLE_STEP_GRAPH=0; Adam / RMSProp / clipping)."""
        # the captured segments read PRIVATE copies of the features: a caller may hand in other tensors (or reuse these) on
        # later steps -- step() copies them into the static buffers -- and must never find its own tensors overwritten
        self._static = _clone_tree(features)
        self._warm_up()
        dot = os.environ.get("RN_GRAPH_DOT")       # tuning aid: the captured segment A as a DOT file (nodes = kernels, edges = dependencies)
        ga = torch.cuda.CUDAGraph(keep_graph=True) if dot else torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread may poll events while the capture is open
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            self._graph_out = self.segment_a(self._static)
        if dot:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
            err = hip.hipGraphDebugDotPrint(ctypes.c_void_p(ga.raw_cuda_graph()), dot.encode(), 0)
            print("hipGraphDebugDotPrint ->", err, dot, flush=True)
            ga.instantiate()
        had_deferred = bool(self._deferred_wgrads) and self.num_parts() > 0      # (forked and joined inside the first part's graph)
        gbs, ranges = [], []
        for j in range(self.num_parts()):          # one graph per part of segment B: the collectives go between the replays
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
                ranges.append(self.segment_b(j))   # (deferred weight gradients, if any, are forked AND joined inside the first part)
            gbs.append(gb)
        self._parts = []
        # (index 6: were tower weight gradients deferred into the first part's graph?  Then the subnets' slice is complete after it)
        self._graphs = (ga, gbs, ranges, self._graph_out, self._static, False, had_deferred)

    def _whole_step_ok(self):
        ar = self.allreduce
        return (self.whole_step_graph and self.use_graph and self.device.type == 'cuda'
                and (not ar.active or (getattr(ar, 'capturable', False) and not getattr(ar, 'host_staged', False)))
                and self.opt.kind == 'momentum' and self.opt.clip <= 0.0 and FUSED_OPT_NORM)

    def _capture_whole(self, features):
        """_run_step as ONE graph (see __init__: whole_step_graph): with several ranks the collectives are nodes of it."""
        self._static = _clone_tree(features)
        self._warm_up()
        g = torch.cuda.CUDAGraph()
        count = self.opt.step_count
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._graph_out = self._run_step(self._static)
        self.opt.step_count = count                 # (recorded, not run: replays count their own steps)
        ranges = [r for r in self.schedule if r[1] <= self.cut_offset]
        # (the parts' arena ranges are kept for reporting; no graph per part)
        self._graphs = (g, [], ranges, self._graph_out, self._static, True, list(self.schedule))

    def step(self, features=None):
        # a feed (dataset.DeviceFeed: input_fn with stage / consumed) puts a NEW sample into its static device buffers before
        # every step -- the captured segment reads them, so the graph path trains on fresh data like the reference's
        # train_input_fn (train.py:190-202); one set of captured segments per input shape
        feed = self.input_fn if (features is None and hasattr(self.input_fn, 'stage')) else None
        key = feed.stage() if feed is not None else None
        whole = self._whole_step_ok()
        if whole:
            # the one-graph step bakes the learning rate into the captured update: a rate that keeps changing (a schedule, a warm-up)
            # would capture again every step -- two warm-up passes + a capture each time, churning the graph cache.  After the third
            # distinct rate the trainer keeps to one graph per part with the update launched eagerly (its scalars are launch arguments).
            if self.opt.lr != self._last_lr:
                self._last_lr, self._lr_changes = self.opt.lr, self._lr_changes + 1
                if self._lr_changes == 4:
                    print("[trainer] the learning rate keeps changing: the update leaves the captured step (one graph per part from here on)",
                          file=sys.stderr, flush=True)
            if self._lr_changes > 3:
                whole = False
        if whole:
            key = (key, 'whole', self.opt.lr, id(self.allreduce))
        if self.use_graph:
            if self._graphs is None:
                self._graph_cache.clear()
            cached = self._graph_cache.get(key)
            if cached is None:
                if self._graph_cache:
                    self.recaptures += 1
                    print("[trainer] new graph key %s (input shape / learning rate / collective): capturing another graph set "
                          "(%d cached, bound %d)" % (key, len(self._graph_cache), self.graph_cache_max), file=sys.stderr, flush=True)
                while len(self._graph_cache) >= self.graph_cache_max:
                    self._graph_cache.popitem(last=False)             # least recently used: its pool is freed with it
                if whole:
                    self._capture_whole(features)
                else:
                    self._capture(features)
                self._graph_cache[key] = self._graphs
            else:
                self._graph_cache.move_to_end(key)
                self._graphs = cached
                self._graph_out, self._static = cached[3], cached[4]
                if features is not None:
                    _copy_tree(self._static, features)
            self._graphs[0].replay()
            class_loss, regr_loss = self._graph_out
            if self._graphs[5]:
                # the whole step was that one replay: what is left is the update's host-side bookkeeping
                self.schedule[:] = self._graphs[6]
                self.opt.step_count += 1
                import ops_f16
                ops_f16.weights_changed()
            else:
                del self.schedule[:]
                ar = self.allreduce

                def reduce(lo, hi):
                    if hi > lo:
                        self.schedule.append((lo, hi))
                        ar.launch(lo, hi)
                deferred = bool(self._graphs[6]) and ar.active
                top = self.heads_offset if deferred else self.arena.count
                reduce(self.cut_offset, top)                          # heads + FPN (or the FPN alone): under the backbone's backward pass
                for j, (gb, (lo, hi)) in enumerate(zip(self._graphs[1], self._graphs[2])):
                    gb.replay()
                    if j == 0 and deferred:
                        reduce(top, self.arena.count)                 # the subnets' slice: its deferred products ran (and joined) inside part 0
                    reduce(lo, hi)                                    # this part's slice: under the parts that follow
                if self.timing is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    grad_scale = ar.wait()
                    e1.record()
                    self.timing.setdefault('exposed_events', []).append((e0, e1))
                else:
                    grad_scale = ar.wait()
                self.opt.step(grad_scale, self.drop_counter)
        else:
            class_loss, regr_loss = self._run_step(features, timed=True)
        if feed is not None:
            feed.consumed()
        self.steps_done += 1
        # (only the opt-in waiting kernels -- the grid-resident GroupNorm, the XCD-resident MobileNetV2 section -- can flag
        # anything; checked after the first steps too, not only every check_interval: a poisoned update must not be trained on
        # for long.  The same check on every path, the one-graph step included.)
        import ops_mb
        if (self.check_interval and (ops.GN_GRID_RESIDENT or ops_mb.RESIDENT)
                and (self.steps_done in (1, 2, 4, 8) or self.steps_done % self.check_interval == 0)):
            self.check_device_errors()
        self.last = {'class_loss': class_loss, 'regr_loss': regr_loss,
                     'regularization_loss': self.opt.regularization_loss}
        return self.last

    def allreduce_exposed_ms(self):
        """Mean time the compute stream waited for collectives after the last backward kernel (needs timing = {})."""
        ev = (self.timing or {}).get('exposed_events', [])
        if not ev:
            return 0.0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in ev) / len(ev))

    def check_device_errors(self):
        """Raise if any kernel of this process has flagged an error on the device (synchronises).  Two opt-in kernel families
        wait for co-resident blocks and can time out: the grid-resident GroupNorm (ops.GN_GRID_RESIDENT) and the XCD-resident
        MobileNetV2 section (ops_mb.RESIDENT); the one that flagged is the one switched off."""
        import ops_mb
        n_gn, n_mb = _rn.barrier_timeout_sources()
        if self.allreduce.active:       # every rank must take the same decision, or the others hang in the next all-reduce
            t = torch.tensor([float(n_gn), float(n_mb)], device=self.device)
            self.allreduce.dist.all_reduce(t, op=self.allreduce.dist.ReduceOp.MAX, group=self.allreduce.group)
            n_gn, n_mb = int(t[0].item()), int(t[1].item())
        if n_gn or n_mb:
            what = []
            if n_gn:
                ops.GN_GRID_RESIDENT = False        # the launch-ordered kernels from here on (on every rank: the counts are maxima over ranks)
                what.append("%d GroupNorm exchange wait(s) (ops.GN_GRID_RESIDENT is now False)" % n_gn)
            if n_mb:
                ops_mb.RESIDENT = False
                what.append("%d MobileNetV2 resident-section barrier(s) (ops_mb.RESIDENT is now False)" % n_mb)
            _rn.reset_barrier_timeouts()
            self._graphs = None                 # the captured graphs contain the waiting kernels: capture again ...
            self._graph_cache.clear()           # ... for every input shape
            raise _rn.RnError("timed out on some rank: %s: the updates since the last check are invalid -- reload the last "
                              "checkpoint and continue" % "; ".join(what))


def _for_each_tensor(tree, fn):
    if torch.is_tensor(tree):
        fn(tree)
    elif isinstance(tree, dict):
        for v in tree.values():
            _for_each_tensor(v, fn)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            _for_each_tensor(v, fn)


def _clone_tree(tree):
    if torch.is_tensor(tree):
        return tree.clone()
    if isinstance(tree, dict):
        return {k: _clone_tree(v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_clone_tree(v) for v in tree)
    return tree


def _copy_tree(dst, src):
    if torch.is_tensor(dst):
        if dst.data_ptr() != src.data_ptr():
            dst.copy_(src)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_tree(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _copy_tree(d, s)


# ------------------------------------------------------------------------------------------------ CLI
# Minimal driver with the reference's flags (train.py:88-108).  Dataset readers (COCO / Pascal, cv2,
# pycocotools) are out of scope; 'shapes' is a synthetic stand-in following data_loaders/shapes.py:133-176
# (1-4 filled squares, half-size in [20, S/4], 3 classes), rendered with numpy.
def evaluate(net, data_loader, levels, num_images, scale=None, device='cuda', score_threshold=0.5):
    """Inference + decode + class-wise NMS (train.py:68-85) over `num_images` samples of the loader, scored with
    COCO-style mAP and the reference's two IoU metrics (train.py:137-161); see metrics.py."""
    import dataset
    import metrics
    dets, gts, ciou, riou = [], [], [], []
    it = dataset.build_dataset(data_loader, levels, scale=scale, device=device)
    with torch.no_grad():
        for _ in range(num_images):
            try:
                b = next(it)
            except StopIteration:
                break
            image = b['image'][:1]                                                 # the un-flipped half
            out = net(image, training=False)
            size = (int(image.shape[1]), int(image.shape[2]))
            anchors = {k: levels[k].normalized_anchor_sizes(size) for k in levels}
            probs = {k: ops.activation(v, 'sigmoid') for k, v in out['classifications'].items()}
            dec = {k: utils.regression_postprocess(out['regressions'][k], anchors[k]) for k in levels}
            d = utils.detect(probs, dec, data_loader.num_classes, score_threshold=score_threshold)[0]
            dets.append((d.boxes.cpu().numpy(), d.scores.cpu().numpy(), d.class_ids.cpu().numpy()))
            gts.append((np.asarray(b['boxes'], np.float32), np.asarray(b['class_ids'])))
            for k in levels:                                                       # build_metrics inputs, per level
                m = b['trainable_masks'][k][:1].cpu().numpy().astype(bool)
                lab = b['detection']['classifications'][k][:1].cpu().numpy()
                ciou.append((lab[m], probs[k].cpu().numpy()[m]))
                fg = lab.max(-1) > 0.5
                true_boxes = utils.regression_postprocess(b['detection']['regressions'][k][:1], anchors[k]).cpu().numpy()
                riou.append((true_boxes[fg], dec[k].cpu().numpy()[fg]))
    res = metrics.mean_average_precision(dets, gts, data_loader.num_classes)
    res['class_iou'] = metrics.class_iou(np.concatenate([a.reshape(-1) for a, _ in ciou]), np.concatenate([p.reshape(-1) for _, p in ciou]))
    res['regr_iou'] = metrics.regr_iou(np.concatenate([a for a, _ in riou]), np.concatenate([p for _, p in riou]))
    res['images'] = len(dets)
    return res


def build_parser():
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument('--learning-rate', type=float, default=1e-2)
    parser.add_argument('--dropout', type=float, default=0.2)
    parser.add_argument('--dataset', type=str, nargs='+', default=['shapes'])
    parser.add_argument('--epochs', type=int, default=1)
    parser.add_argument('--scale', type=int, default=256)
    parser.add_argument('--experiment', type=str, default=None, help='directory for checkpoints (model.safetensors)')
    parser.add_argument('--grad-clip-norm', type=float)
    parser.add_argument('--backbone', type=str, choices=['resnet_50', 'densenet_121', 'densenet_169', 'mobilenet_v2'],
                        default='mobilenet_v2')
    parser.add_argument('--optimizer', type=str, choices=['momentum', 'adam', 'rmsprop'], default='momentum')
    parser.add_argument('--loss', type=str, choices=['bce_dice', 'focal'], default='bce_dice')
    parser.add_argument('--steps-per-epoch', type=int, default=100)
    parser.add_argument('--eval-images', type=int, default=0, help='after training: mAP / IoU metrics over this many samples')
    parser.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying the captured step')
    return parser


def init_distributed(backend=None):
    """One process per GPU, like the reference picks MirroredStrategy from the number of GPUs (train.py:261-267):
    rank / world size / device come from the launcher's environment (python -m torch.distributed.run, or
    bench.py --gpus N which starts that itself).  Returns (device, rank, world, process_group_initialised).
    Must run before anything else touches the GPU: it selects the device every kernel launch then uses."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise _rn.RnError("training needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    started = False
    if 'WORLD_SIZE' in os.environ and 'RANK' in os.environ:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if not dist.is_initialized():
            dist.init_process_group(backend or 'nccl', rank=rank, world_size=world,
                                    **({'device_id': dev} if (backend or 'nccl') == 'nccl' else {}))
            started = True
    return dev, rank, world, started


def broadcast_initial_state(trainer, src=0):
    """Replicas must start from identical variables (MirroredStrategy creates them once and mirrors them)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(trainer.arena.weights, src=src)
        dist.broadcast(trainer.opt.state1, src=src)
        if trainer.opt.state2 is not None:
            dist.broadcast(trainer.opt.state2, src=src)


def main(argv=None):
    import checkpoint
    import dataset
    import retinanet
    args = build_parser().parse_args(argv)
    assert args.dataset[0] == 'shapes', 'only the synthetic shapes loader is built in (file readers are out of scope)'
    dev, rank, world, started = init_distributed()
    from data_loaders.shapes import Shapes
    # every replica draws its own samples (dataset.py:182-204: a replica's batch is [sample, hflip(sample)])
    loader = Shapes(None, image_size=(args.scale + args.scale // 4, args.scale), seed=rank)   # rescale_image brings it to --scale
    levels = build_levels()
    torch.manual_seed(0)                                                           # same initial weights on every rank
    net = retinanet.RetinaNet(backbone=args.backbone, levels=levels, num_classes=loader.num_classes, activation=L.elu,
                              dropout_rate=args.dropout).to(dev)
    # train_input_fn (train.py:190-202) = dataset.DeviceFeed: loader thread -> pinned host memory -> asynchronous upload ->
    # static device buffers; rescale, normalisation, the h-flip and the anchor assignment of every NEW sample run inside the
    # captured step (one hipGraph per backward segment, replayed per step; --no-graph launches the same kernels eagerly)
    feed = dataset.DeviceFeed(loader, levels, scale=args.scale, device=dev)
    trainer = Trainer(net, levels, optimizer=args.optimizer, learning_rate=args.learning_rate,
                      grad_clip_norm=args.grad_clip_norm, loss_mode=args.loss, device=dev, use_graph=not args.no_graph,
                      input_fn=feed)
    step = 0
    drawn0 = 0
    path = None if args.experiment is None else os.path.join(args.experiment, 'model.safetensors')
    if path is not None and os.path.exists(path):
        step = checkpoint.load(path, net, trainer)                                 # every rank reads the same file
        # like the reference (train.py:271-273: `for epoch in range(args.epochs): estimator.train(...)` on a restored global
        # step) a rerun trains args.epochs MORE epochs; what is carried over besides the weights: the step count, the
        # optimizer slots, the dropout counter and the position in the sample stream (counted in samples, not steps)
        drawn0 = int(checkpoint.load_extra(path).get('samples_drawn', 0))
        if rank == 0:
            print('restored step', step)
    loader.skip(drawn0)                                                            # (before the feed's thread draws: it starts lazily)
    broadcast_initial_state(trainer)
    feed.start()
    try:
        for epoch in range(args.epochs):
            for _ in range(args.steps_per_epoch):
                out = trainer.step()                                               # batch = [image, hflip] of a new sample
                step += 1
                if step % 20 == 0 and rank == 0:
                    print('epoch %d step %d class_loss %.4f regr_loss %.4f reg %.4f' % (
                        epoch, step, out['class_loss'].item(), out['regr_loss'].item(), out['regularization_loss'].item()),
                        flush=True)
            trainer.check_device_errors()
            if path is not None and rank == 0:                                     # replicas are identical: rank 0 writes
                checkpoint.save(path, net, trainer, step=step,
                                extra={'epochs_done': epoch + 1, 'samples_drawn': drawn0 + feed.samples_staged})
    finally:
        feed.close()
    if args.eval_images and rank == 0:
        res = evaluate(net, Shapes(None, image_size=(args.scale + args.scale // 4, args.scale), seed=12345), levels,
                       args.eval_images, scale=args.scale, device=dev)
        print('eval: mAP %.4f AP50 %.4f AP75 %.4f class_iou %.4f regr_iou %.4f over %d images' % (
            res['mAP'], res['AP50'], res['AP75'], res['class_iou'], res['regr_iou'], res['images']), flush=True)
    if started:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return step


if __name__ == '__main__':
    main()
