"""Autograd carriers for the gfx950 kernels (librn_hip.so).

Each ``torch.autograd.Function`` here only allocates outputs, fills the C-ABI segment
structs with raw pointers and launches the HIP kernels on the current stream; all the
arithmetic is in ``csrc/``.  Lists of tensors are "segments" that share the layer's
parameters and run in one launch (the shared heads over P3..P7, reference
retinanet.py:283-291).
"""
import ctypes as C
import os

import torch

import _rn


# When True (set by train.Trainer), parameter gradients are written by the kernels straight into
# the parameter's pre-allocated `.grad` (a view of the flat gradient arena) and autograd is handed
# None for them: no per-parameter accumulate kernels, no arena memset.  Requires every parameter
# to be used by exactly one op call per backward pass (true for this network: shared heads are ONE
# multi-segment call).
DIRECT_PARAM_GRADS = False


# Weight gradients are off the critical path of backward (only the optimizer consumes them) while the
# dgrad -> GroupNorm-backward chain is a long sequence of small, latency-bound kernels: run wgrad (+ its
# slab reduce and the bias gradient) on a side stream so the matrix cores work on it underneath that
# chain.  A replayed hipGraph keeps the two branches concurrent (tools/graph_concurrency.py).  The
# consumer (train.Trainer / any caller that reads .grad) must call _rn.join_side_streams() first --
# torch.autograd.backward's own end-of-backward sync only covers streams it knows about.
WGRAD_SIDE_STREAM = False


class _on_side_stream(object):
    """with _on_side_stream(device, tensors): ...  -- fork after the current stream, keep `tensors` alive
    for the side stream (caching-allocator record_stream)."""

    def __init__(self, device, tensors, direct=True):
        # only when the results go straight into pre-allocated .grad buffers: a tensor handed back to
        # autograd would be consumed on the main stream without waiting for the side stream
        self.on = WGRAD_SIDE_STREAM and direct and device.type == 'cuda'
        if self.on:
            self.side = _rn.side_stream(device, 0)
            self.side.wait_stream(torch.cuda.current_stream(device))
            for t in tensors:
                if t is not None:
                    t.record_stream(self.side)
            self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        if self.on:
            self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.on:
            self.ctx.__exit__(*a)
        return False


# Winograd F(4x4,3x3) (or F(2x2,3x3): WINOGRAD_TILE = 2) for dense 3x3 / stride-1 convs wide enough to pay for
# the transforms (head towers, FPN merges, the 720-wide class output conv): forward, data gradient and (WINOGRAD_WGRAD)
# weight gradient.
WINOGRAD = True
WINOGRAD_TILE = 4
WINOGRAD_WGRAD = True
MERGED_CONV_BWD = True    # data + weight gradient of a small conv in one launch (rn_conv2d_bwd)
WINOGRAD_KEEP = True      # forward keeps the transformed input / rotated kernel for backward (a few hundred MB per step)
WINOGRAD_MIN_CHANNELS = 64
WINOGRAD_MAX_WORKSPACE = int(os.environ.get("RN_WINOGRAD_MAX_WS", 8 << 30))   # of 288 GB HBM; the V / M planes of 1024^2 x 16 need 1.7 GB


def _winograd_ok(w, stride, groups, xs):
    kh, kw, cin, cout = w.shape
    if not (WINOGRAD and kh == 3 and kw == 3 and stride == 1 and groups == 1):
        return False
    if cin % 4 or cout % 4 or min(cin, cout) < WINOGRAD_MIN_CHANNELS:
        return False
    m = WINOGRAD_TILE
    tiles = sum(x.shape[0] * ((x.shape[1] + m - 1) // m) * ((x.shape[2] + m - 1) // m) for x in xs)
    return 4 * (m + 2) ** 2 * (tiles * (cin + cout) + cin * cout) <= WINOGRAD_MAX_WORKSPACE


def _winograd(segs, n, w, bias, dgrad, v_buf=None, urot_buf=None):
    L = _rn.lib()
    cin, cout = w.shape[2], w.shape[3]
    need = L.rn_conv3x3_winograd_workspace(segs, n, cin, cout, WINOGRAD_TILE)
    ws = _rn.workspace(need, w.device)
    _rn.check(L.rn_conv3x3_winograd(segs, n, cin, cout, _rn.f32(w), _rn.f32(bias) if bias is not None else None,
                                    1 if dgrad else 0, WINOGRAD_TILE, ws.data_ptr(), ws.numel(),
                                    _rn.f32(v_buf) if v_buf is not None else None,
                                    _rn.f32(urot_buf) if urot_buf is not None else None, _rn.stream()),
              "rn_conv3x3_winograd")


def _winograd_keep_buffers(segs, n, w, want_v, want_urot):
    """Buffers a training-mode forward call fills for its backward pass (see rn_conv3x3_winograd)."""
    if not WINOGRAD_KEEP or not (want_v or want_urot):
        return None, None
    vb, ub = C.c_size_t(0), C.c_size_t(0)
    _rn.check(_rn.lib().rn_conv3x3_winograd_keep_bytes(segs, n, w.shape[2], w.shape[3], WINOGRAD_TILE, C.byref(vb), C.byref(ub)),
              "rn_conv3x3_winograd_keep_bytes")
    v = torch.empty((vb.value // 4,), dtype=torch.float32, device=w.device) if want_v else None
    u = torch.empty((ub.value // 4,), dtype=torch.float32, device=w.device) if want_urot else None
    return v, u


# GroupNorm partial sums as a by-product of the producing conv / depthwise kernel (rn_conv2d_fwd_stats, rn_depthwise_fwd_stats):
# the GroupNorm that follows merges the rows while its activations load, and reads x once.  model.Sequential announces
# the GroupNorm to the conv in front of it; the rows travel on the conv's output tensor (`_gn_rows`).
GN_PRODUCER_STATS = os.environ.get("RN_GN_PRODUCER_STATS", "1") == "1"
_LAST_ROWS_LAYOUT = [None]      # (rows_per_sample, per_group, groups) of the rows the last producer call returned


def _conv_fwd(segs, n, geom, device, gn=None, samples=0):
    """The forward conv; with `gn` = (groups, eps) of a GroupNorm that follows, and a shape whose kernel can emit them, also the
    partial-sum rows of the output: returns (rows tensor, rows_per_sample, per_group, groups) or None."""
    L = _rn.lib()
    need = L.rn_conv2d_fwd_workspace(segs, n, C.byref(geom))           # split-K scratch for tiny grids (0 otherwise)
    if gn is not None and GN_PRODUCER_STATS and device.type == 'cuda':
        g = gn_groups(segs[0].cout, gn[0])
        lay = _rn.GnRows(None, 0, 0, g)
        nbytes = L.rn_conv2d_stats_rows(segs, n, C.byref(geom), need, g, C.byref(lay))
        if nbytes:
            rows = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
            lay.rows = rows.data_ptr()
            # (the scratch rn_conv2d_stats_rows was told about: the patch matrix of a strided dense conv on the split-bf16 kernels)
            ws = _rn.workspace(need, device) if need else None
            _rn.check(L.rn_conv2d_fwd_stats(segs, n, C.byref(geom), ws.data_ptr() if need else None, ws.numel() if need else 0, C.byref(lay),
                                            _rn.stream()), "rn_conv2d_fwd_stats")
            return rows, lay.rows_per_sample, lay.per_group, g
    ws = _rn.workspace(need, device) if need else None
    _rn.check(L.rn_conv2d_fwd(segs, n, C.byref(geom), ws.data_ptr() if need else None, ws.numel() if need else 0,
                              _rn.stream()), "rn_conv2d_fwd")
    return None


def _conv_dgrad(segs, n, geom, device):
    L = _rn.lib()
    need = L.rn_conv2d_dgrad_workspace(segs, n, C.byref(geom))
    ws = _rn.workspace(need, device) if need else None
    _rn.check(L.rn_conv2d_dgrad(segs, n, C.byref(geom), ws.data_ptr() if need else None, ws.numel() if need else 0,
                                _rn.stream()), "rn_conv2d_dgrad")


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


_direct_written = set()      # ids of the parameters whose gradient slot has been written since begin_direct_grad_step()


def begin_direct_grad_step():
    """Start of a training step under DIRECT_PARAM_GRADS (train.Trainer calls it): forget which slots were written."""
    _direct_written.clear()


def _grad_slot(p):
    """(buffer to write the gradient of parameter p into, value to return to autograd).  Under DIRECT_PARAM_GRADS the kernels
    OVERWRITE p.grad in place and autograd gets None: that is only right while every parameter is used by exactly one op
    call per step, so a second write within a step is refused instead of silently dropping the first contribution."""
    if DIRECT_PARAM_GRADS and p.grad is not None and p.grad.is_contiguous():
        if id(p) in _direct_written:
            raise RuntimeError("DIRECT_PARAM_GRADS: a parameter of shape %s is used by two op calls in one step (shared weights must "
                               "go through ONE call, e.g. a list of inputs); its first gradient would be overwritten" % (tuple(p.shape),))
        _direct_written.add(id(p))
        return p.grad, None
    assert not _deferring, "deferred reductions need every parameter gradient written in place (DIRECT_PARAM_GRADS)"
    g = torch.empty_like(p)
    return g, g


# Deferred gradient reductions: between begin_deferred_reductions() and end_deferred_reductions() the fixed-order row
# reductions that finish every weight / bias / GroupNorm-parameter gradient are recorded in a caller-owned rn_reduce_list
# per stream (handed to each gradient entry point as its `defer` argument -- the library keeps nothing) and run as ONE
# launch per stream at the end instead of ~130 launch-latency-bound kernels per step.  Their inputs (per-call workspaces)
# are kept alive here until the flush.
_deferring = False
_deferred_keep = []
_deferred_streams = []
_defer_lists = {}          # raw stream handle -> _rn.ReduceList (allocated once, reused every step)


def _defer_arg():
    """The `defer` argument for an entry point launched on the current stream: this stream's list while deferring."""
    if not _deferring:
        return None
    lst = _defer_lists.get(torch.cuda.current_stream().cuda_stream)
    return C.byref(lst) if lst is not None else None


def begin_deferred_reductions(extra_streams=()):
    """Defer on the current stream and on `extra_streams` (torch streams that backward nodes run on, e.g. the
    second head stream of retinanet.HEADS_TWO_STREAMS)."""
    global _deferring
    assert DIRECT_PARAM_GRADS, "deferred reductions write parameter gradients in place: set DIRECT_PARAM_GRADS"
    for h in [torch.cuda.current_stream().cuda_stream] + [st.cuda_stream for st in extra_streams]:
        lst = _defer_lists.get(h)
        if lst is None:
            lst = _defer_lists[h] = _rn.ReduceList.make()
        lst.count = 0
    _deferred_streams[:] = list(extra_streams)
    _deferring = True


def end_deferred_reductions():
    """Flush: one batched reduction launch per stream; the current stream then waits for the extra streams."""
    global _deferring
    if _deferring:
        L = _rn.lib()
        _deferring = False
        cur = torch.cuda.current_stream()
        try:
            _rn.check(L.rn_flush_reductions(C.byref(_defer_lists[cur.cuda_stream]), _rn.stream()), "rn_flush_reductions")
            for st in _midway_streams:             # the flushes started midway (their inputs stay alive in _deferred_keep until here)
                cur.wait_stream(st)
            del _midway_streams[:]
            for st in _deferred_streams:
                # fork again from the current stream first: after autograd's end-of-backward join a side stream is no
                # longer part of an ongoing hipGraph capture, and its flush would run eagerly instead of being captured
                st.wait_stream(cur)
                _rn.check(L.rn_flush_reductions(C.byref(_defer_lists[st.cuda_stream]), C.c_void_p(st.cuda_stream)),
                          "rn_flush_reductions")
                cur.wait_stream(st)
        finally:
            del _deferred_keep[:]
            del _deferred_streams[:]


FLUSH_MIDWAY = os.environ.get("RN_FLUSH_MIDWAY", "0") == "1"     # measured: 446 vs 450 images/s -- off (a fork / join edge costs more than the tail it shortens)


def flush_deferred_midway(device):
    """Run the row reductions recorded so far on the current stream NOW, on a side stream (index 4), and go on recording: a
    long backward node (the MobileNetV2 chain: ~50 reductions, most of them recorded while the small maps are processed) then
    leaves only its last few for the flush that sits on the critical path after the last backward kernel.  The side stream is
    joined by end_deferred_reductions (and by _rn.join_side_streams)."""
    if not (_deferring and FLUSH_MIDWAY):
        return
    cur = torch.cuda.current_stream()
    lst = _defer_lists.get(cur.cuda_stream)
    if lst is None or lst.count == 0:
        return
    side = _rn.side_stream(device, 4)
    side.wait_stream(cur)
    _rn.check(_rn.lib().rn_flush_reductions(C.byref(lst), C.c_void_p(side.cuda_stream)), "rn_flush_reductions")
    _midway_streams.append(side)


_midway_streams = []


def _grad_workspace(need, device):
    """Workspace of a gradient kernel whose last stage is a row reduction: private while deferring."""
    if not _deferring:
        return _rn.workspace(need, device)
    ws = torch.empty((max(int(need), 256),), dtype=torch.uint8, device=device)
    _deferred_keep.append(ws)
    return ws


def _conv_segs(xs, w, bias, ys, dys, dxs, x_ld=0, x_coff=0):
    cout = w.shape[3]
    segs = (_rn.ConvSeg * len(xs))()
    for i, x in enumerate(xs):
        s = segs[i]
        s.x_ld, s.x_coff = x_ld, x_coff
        s.x = _rn.f32(x) if x is not None else None
        s.wgt = _rn.f32(w)
        s.bias = _rn.f32(bias) if bias is not None else None
        s.y = _rn.f32(ys[i]) if ys is not None else None
        s.dy = _rn.f32(dys[i]) if dys is not None else None
        s.dx = _rn.f32(dxs[i]) if dxs is not None else None
        s.n, s.h, s.w = x.shape[0], x.shape[1], x.shape[2]
        s.cout = cout
    return segs


class _Conv2dShared(torch.autograd.Function):
    """tf.layers.Conv2D(padding='same') applied to n inputs that share one kernel."""

    @staticmethod
    def forward(ctx, stride_groups, w, bias, *xs):
        stride, groups = stride_groups[:2]
        gn = stride_groups[2] if len(stride_groups) > 2 else None      # (groups, eps) of the GroupNorm that follows
        kh, kw, cin_g, cout = w.shape
        cin = cin_g * groups
        L = _rn.lib()
        geom = _rn.ConvGeom(kh, kw, stride, cin, groups)
        ys = []
        for x in xs:
            assert x.dim() == 4 and x.shape[3] == cin, "conv2d: input %s vs kernel %s" % (tuple(x.shape), tuple(w.shape))
            oh, _ = _rn.same_pad(x.shape[1], kh, stride)
            ow, _ = _rn.same_pad(x.shape[2], kw, stride)
            ys.append(torch.empty((x.shape[0], oh, ow, cout), dtype=torch.float32, device=x.device))
        xs = [x.contiguous() for x in xs]
        segs = _conv_segs(xs, w, bias, ys, None, None)
        ctx.winograd = _winograd_ok(w, stride, groups, xs)
        ctx.wino_v = ctx.wino_urot = None
        if ctx.winograd:
            # (grad mode is always off inside Function.forward: ctx.needs_input_grad is what says "training")
            if any(ctx.needs_input_grad):
                ctx.wino_v, ctx.wino_urot = _winograd_keep_buffers(
                    segs, len(xs), w, ctx.needs_input_grad[1] and WINOGRAD_WGRAD, any(ctx.needs_input_grad[3:]))
            _winograd(segs, len(xs), w, bias, False, ctx.wino_v, ctx.wino_urot)
            stats = None
        else:
            stats = _conv_fwd(segs, len(xs), geom, xs[0].device, gn if (gn and len(xs) == 1 and bias is None) else None, xs[0].shape[0])
        ctx.stride = stride
        ctx.groups = groups
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.save_for_backward(w, *xs)
        ctx.set_materialize_grads(False)
        if stats is not None:      # the partial-sum rows of y came with it: one more (non-differentiable) output
            _LAST_ROWS_LAYOUT[0] = tuple(stats[1:])
            ctx.mark_non_differentiable(stats[0])
            return tuple(ys) + (stats[0],)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        w, *xs = ctx.saved_tensors
        bias = ctx.bias_ref
        kh, kw, cin_g, cout = w.shape
        cin = cin_g * ctx.groups
        L = _rn.lib()
        geom = _rn.ConvGeom(kh, kw, ctx.stride, cin, ctx.groups)
        n = len(xs)
        dys = [dy.contiguous() if dy is not None else None for dy in dys[:n]]
        for i in range(n):
            if dys[i] is None:
                oh, _ = _rn.same_pad(xs[i].shape[1], kh, ctx.stride)
                ow, _ = _rn.same_pad(xs[i].shape[2], kw, ctx.stride)
                dys[i] = torch.zeros((xs[i].shape[0], oh, ow, cout), dtype=torch.float32, device=w.device)
        need_dx = [ctx.needs_input_grad[3 + i] for i in range(n)]
        dxs = [None] * n
        want_dw0 = ctx.needs_input_grad[1]
        if all(need_dx) and want_dw0 and ctx.winograd and WINOGRAD_WGRAD and not WGRAD_SIDE_STREAM and MERGED_CONV_BWD:
            # the whole backward pass of a Winograd layer in three launches (transforms, products, back-transforms)
            outs = [torch.empty_like(x) for x in xs]
            segs = _conv_segs(xs, w, None, None, dys, outs)
            dw_buf, dw = _grad_slot(w)
            have_v, have_u = ctx.wino_v is not None, ctx.wino_urot is not None
            need = L.rn_conv3x3_winograd_bwd_workspace(segs, n, cin, cout, WINOGRAD_TILE, 1 if have_v else 0, 1 if have_u else 0)
            ws = _rn.workspace(need, w.device)
            _rn.check(L.rn_conv3x3_winograd_bwd(segs, n, cin, cout, _rn.f32(w), _rn.f32(dw_buf), 0, WINOGRAD_TILE, ws.data_ptr(),
                                                ws.numel(), _rn.f32(ctx.wino_v) if have_v else None,
                                                _rn.f32(ctx.wino_urot) if have_u else None, _rn.stream()), "rn_conv3x3_winograd_bwd")
            db = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db_buf, db = _grad_slot(bias)
                need = L.rn_conv2d_bias_grad_workspace(cout)
                ws = _grad_workspace(need, w.device)
                _rn.check(L.rn_conv2d_bias_grad(segs, n, C.byref(geom), _rn.f32(db_buf), ws.data_ptr(), ws.numel(), _rn.stream(), _defer_arg()),
                          "rn_conv2d_bias_grad")
            return (None, dw, db) + tuple(outs)
        if (all(need_dx) and want_dw0 and not ctx.winograd and not WGRAD_SIDE_STREAM and n <= 4 and MERGED_CONV_BWD):
            # both gradients from one launch (small convs are launch-latency-bound); not when dgrad would take split-K
            outs = [torch.empty_like(x) for x in xs]
            segs = _conv_segs(xs, w, None, None, dys, outs)
            if L.rn_conv2d_dgrad_workspace(segs, n, C.byref(geom)) == 0:
                dw_buf, dw = _grad_slot(w)
                need = L.rn_conv2d_wgrad_workspace(segs, n, C.byref(geom))
                ws = _grad_workspace(need, w.device)
                _rn.check(L.rn_conv2d_bwd(segs, n, C.byref(geom), _rn.f32(dw_buf), ws.data_ptr(), ws.numel(), _rn.stream(), _defer_arg()),
                          "rn_conv2d_bwd")
                db = None
                if ctx.has_bias and ctx.needs_input_grad[2]:
                    db_buf, db = _grad_slot(bias)
                    need = L.rn_conv2d_bias_grad_workspace(cout)
                    ws = _grad_workspace(need, w.device)
                    _rn.check(L.rn_conv2d_bias_grad(segs, n, C.byref(geom), _rn.f32(db_buf), ws.data_ptr(), ws.numel(),
                                                    _rn.stream(), _defer_arg()), "rn_conv2d_bias_grad")
                return (None, dw, db) + tuple(outs)
        if any(need_dx):
            idx = [i for i in range(n) if need_dx[i]]
            outs = [torch.empty_like(xs[i]) for i in idx]
            segs = _conv_segs([xs[i] for i in idx], w, None, None, [dys[i] for i in idx], outs)
            if ctx.winograd:
                _winograd(segs, len(idx), w, None, True, None, ctx.wino_urot)
            else:
                _conv_dgrad(segs, len(idx), geom, w.device)
            for i, o in zip(idx, outs):
                dxs[i] = o
        dw = db = None
        want_dw = ctx.needs_input_grad[1]
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if want_dw or want_db:
            dw_buf = db_buf = None
            if want_dw:
                dw_buf, dw = _grad_slot(w)
            if want_db:
                db_buf, db = _grad_slot(bias)
            with _on_side_stream(w.device, list(xs) + list(dys) + [dw_buf, db_buf], direct=(dw is None and db is None)):
                segs = _conv_segs(xs, w, None, None, dys, None)
                if want_dw and ctx.winograd and WINOGRAD_WGRAD:
                    need = L.rn_conv3x3_winograd_wgrad_workspace(segs, n, cin, cout, WINOGRAD_TILE)
                    ws = _rn.workspace(need, w.device)
                    _rn.check(L.rn_conv3x3_winograd_wgrad(segs, n, cin, cout, _rn.f32(dw_buf), 0, WINOGRAD_TILE, ws.data_ptr(),
                                                          ws.numel(), _rn.f32(ctx.wino_v) if ctx.wino_v is not None else None,
                                                          _rn.stream()), "rn_conv3x3_winograd_wgrad")
                elif want_dw:
                    need = L.rn_conv2d_wgrad_workspace(segs, n, C.byref(geom))
                    ws = _grad_workspace(need, w.device)
                    _rn.check(L.rn_conv2d_wgrad(segs, n, C.byref(geom), _rn.f32(dw_buf), 0, ws.data_ptr(), ws.numel(),
                                                _rn.stream(), _defer_arg()), "rn_conv2d_wgrad")
                if want_db:
                    need = L.rn_conv2d_bias_grad_workspace(cout)
                    ws = _grad_workspace(need, w.device)
                    _rn.check(L.rn_conv2d_bias_grad(segs, n, C.byref(geom), _rn.f32(db_buf), ws.data_ptr(), ws.numel(),
                                                    _rn.stream(), _defer_arg()), "rn_conv2d_bias_grad")
        return (None, dw, db) + tuple(dxs)


class _Conv2dChannelSplit(torch.autograd.Function):
    """k convs (own kernel + bias each) that read consecutive channel slices of the SAME input tensors:
    slice j = channels [off_j, off_j + cin_j).  No slice copies: the kernels take the pixel stride and
    channel offset (rn_conv_seg.x_ld / x_coff); in backward each conv's dgrad fills its slice of one
    shared dx buffer.  Used for the class / box output convs on the fused 512-channel head towers."""

    @staticmethod
    def forward(ctx, stride, k, *args):
        ws, bs, xs = list(args[:k]), list(args[k:2 * k]), [x.contiguous() for x in args[2 * k:]]
        L = _rn.lib()
        ld = xs[0].shape[3]
        outs, off, offs = [], 0, []
        for w, b in zip(ws, bs):
            kh, kw, cin, cout = w.shape
            geom = _rn.ConvGeom(kh, kw, stride, cin, 1)
            ys = []
            for x in xs:
                oh, _ = _rn.same_pad(x.shape[1], kh, stride)
                ow, _ = _rn.same_pad(x.shape[2], kw, stride)
                ys.append(torch.empty((x.shape[0], oh, ow, cout), dtype=torch.float32, device=x.device))
            segs = _conv_segs(xs, w, b, ys, None, None, ld, off)
            _conv_fwd(segs, len(xs), geom, xs[0].device)
            outs += ys
            offs.append(off)
            off += cin
        assert off == ld, "channel slices must cover the input exactly"
        ctx.cfg = (stride, k, offs, bs)
        ctx.save_for_backward(*ws, *xs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        stride, k, offs, bs = ctx.cfg
        saved = ctx.saved_tensors
        ws, xs = list(saved[:k]), list(saved[k:])
        n = len(xs)
        ld = xs[0].shape[3]
        L = _rn.lib()
        dev = xs[0].device
        dxs = [torch.empty_like(x) for x in xs]
        dws, dbs = [], []
        for j, w in enumerate(ws):
            kh, kw, cin, cout = w.shape
            geom = _rn.ConvGeom(kh, kw, stride, cin, 1)
            dyj = [dy.contiguous() for dy in dys[j * n:(j + 1) * n]]
            segs = _conv_segs(xs, w, None, None, dyj, dxs, ld, offs[j])
            _rn.check(L.rn_conv2d_dgrad(segs, n, C.byref(geom), None, 0, _rn.stream()), "rn_conv2d_dgrad")
            need = L.rn_conv2d_wgrad_workspace(segs, n, C.byref(geom))
            ws_buf = _grad_workspace(need, dev)
            dw_buf, dw = _grad_slot(w)
            _rn.check(L.rn_conv2d_wgrad(segs, n, C.byref(geom), _rn.f32(dw_buf), 0, ws_buf.data_ptr(), ws_buf.numel(),
                                        _rn.stream(), _defer_arg()), "rn_conv2d_wgrad")
            dws.append(dw)
            db = None
            if bs[j] is not None:
                need = L.rn_conv2d_bias_grad_workspace(cout)
                ws_buf = _grad_workspace(need, dev)
                db_buf, db = _grad_slot(bs[j])
                _rn.check(L.rn_conv2d_bias_grad(segs, n, C.byref(geom), _rn.f32(db_buf), ws_buf.data_ptr(),
                                                ws_buf.numel(), _rn.stream(), _defer_arg()), "rn_conv2d_bias_grad")
            dbs.append(db)
        return (None, None) + tuple(dws) + tuple(dbs) + tuple(dxs)


def conv2d_channel_split(xs, weights, biases, stride=1):
    """[conv_j(x[..., off_j:off_j+cin_j]) for j] for every x in xs -> list (per conv) of lists (per x)."""
    k, n = len(weights), len(xs)
    out = _Conv2dChannelSplit.apply(stride, k, *weights, *biases, *xs)
    return [list(out[j * n:(j + 1) * n]) for j in range(k)]


def conv2d(x, w, bias=None, stride=1, groups=1, gn=None):
    """NHWC conv, HWIO kernel [kh,kw,cin/groups,cout], TF SAME padding.  `x` may be a list (shared
    kernel, one launch).  groups > 1: grouped conv (ResNeXt cardinality).  gn = (groups, eps) announces the GroupNorm that
    follows: where the kernel can, it also emits that GroupNorm's statistics (they travel on the returned tensor)."""
    if isinstance(x, (list, tuple)):
        return list(_Conv2dShared.apply((stride, groups), w, bias, *x))
    if gn is not None and GN_PRODUCER_STATS and x.is_cuda and x.dtype == torch.float32:
        outs = _Conv2dShared.apply((stride, groups, (int(gn[0]), float(gn[1]))), w, bias, x)
        y = outs[0]
        if len(outs) == 2:         # group_norm_act finds the rows on the tensor it is handed
            y._gn_rows = (outs[1],) + _LAST_ROWS_LAYOUT[0]
        return y
    return _Conv2dShared.apply((stride, groups), w, bias, x)[0]


# GroupNorm folded into chains of Winograd layers (rn_conv3x3_winograd_gn): the head towers run without GroupNorm kernels
WINO_GN_FOLD = os.environ.get("RN_WINO_GN_FOLD", "1") == "1"


def wino_tower_ok(xs, tower, out_w, groups):
    """Can [conv3x3, GroupNorm, act] x k (+ an output conv3x3) run as folded Winograd layers?  `tower`: [(w, gamma, beta)]."""
    if not (WINO_GN_FOLD and WINOGRAD and xs and all(x.is_cuda and x.dtype == torch.float32 for x in xs)):
        return False
    cin = xs[0].shape[3]
    for w, gamma, beta in tower:
        kh, kw, ci, co = w.shape
        g = gn_groups(co, groups)
        if not (kh == 3 and kw == 3 and ci == cin and ci % 64 == 0 and co % 64 == 0 and 64 % (co // g) == 0):
            return False
        cin = co
    if out_w is not None and not (out_w.shape[0] == 3 and out_w.shape[1] == 3 and out_w.shape[2] == cin and out_w.shape[3] % 4 == 0):
        return False
    m = WINOGRAD_TILE
    tiles = sum(x.shape[0] * ((x.shape[1] + m - 1) // m) * ((x.shape[2] + m - 1) // m) for x in xs)
    widest = max([cin] + [t[0].shape[3] for t in tower] + ([out_w.shape[3]] if out_w is not None else []))
    return 4 * (m + 2) ** 2 * (tiles * 2 * widest + widest * widest) <= WINOGRAD_MAX_WORKSPACE


# Kernel transforms prepared ahead of the layers (rn_conv3x3_winograd_gn_weights; train.Trainer with RN_WINO_PRE=1, opt-in: measured
# slower than transforming inside the layers' first launch, see train.Trainer.wino_pre): the layers' own launches then carry no
# kernel-transform blocks.  {w.data_ptr(): (u_buf, urot_buf, event)}; set for the duration of ONE forward pass.
WINO_PRE = {}


class WinoPretransform(object):
    """Persistent U / Urot buffers of a list of conv kernels [3,3,cin,cout] and the launches that fill them."""

    def __init__(self, weights):
        L = _rn.lib()
        self.items = []
        for w in weights:
            cin, cout = int(w.shape[2]), int(w.shape[3])
            ub = int(L.rn_conv3x3_winograd_gn_u_bytes(cin, cout, WINOGRAD_TILE))
            u = torch.empty((ub + 3) // 4, dtype=torch.float32, device=w.device)
            urot = torch.empty((WINOGRAD_TILE + 2) ** 2 * cin * cout, dtype=torch.float32, device=w.device)
            self.items.append((w, u, urot))

    def launch(self):
        """The transforms on the CURRENT stream; returns the registry for ops.WINO_PRE (the event: recorded behind them)."""
        L = _rn.lib()
        for w, u, urot in self.items:
            _rn.check(L.rn_conv3x3_winograd_gn_weights(_rn.f32(w), int(w.shape[2]), int(w.shape[3]), WINOGRAD_TILE, u.data_ptr(), u.numel() * 4,
                                                       _rn.f32(urot), _rn.stream()), "rn_conv3x3_winograd_gn_weights")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return {w.data_ptr(): (u, urot, ev) for w, u, urot in self.items}, ev


class _WinoTower(torch.autograd.Function):
    """k x [conv3x3 (no bias) -> GroupNorm -> activation] (+ optionally a final conv3x3 with bias) on n tensors that share
    every parameter, as k (+1) folded Winograd layers: see rn_conv3x3_winograd_gn.  Without the final conv the last
    GroupNorm is NOT part of the node: it returns the raw output of conv k (a plain tensor with a plain gradient)."""

    @staticmethod
    def forward(ctx, cfg, n, *args):
        groups, eps, act, k, with_out = cfg
        xs = [x.contiguous() for x in args[:n]]
        params = list(args[n:])
        tower = [(params[3 * i], params[3 * i + 1], params[3 * i + 2]) for i in range(k)]
        out_w, out_b = (params[3 * k], params[3 * k + 1]) if with_out else (None, None)
        L = _rn.lib()
        dev = xs[0].device
        tile = WINOGRAD_TILE
        training = any(ctx.needs_input_grad)
        convs = [(w, None) for (w, _, _) in tower] + ([(out_w, out_b)] if with_out else [])
        nfold = k if with_out else k - 1               # GroupNorms folded between two convs of this node
        cur, ys_all, rows_all, keep = xs, [], [], []
        nrows = None
        for i, (w, b) in enumerate(convs):
            cin, cout = w.shape[2], w.shape[3]
            ys = [torch.empty((x.shape[0], x.shape[1], x.shape[2], cout), dtype=torch.float32, device=dev) for x in cur]
            segs = _conv_segs(cur, w, b, ys, None, None)
            if nrows is None:
                nrows = int(L.rn_wino_gn_rows(segs, n, tile))
            gn = _rn.WinoGn()
            if i > 0:                                   # the input is the raw output of conv i-1: normalise while loading
                _, gamma, beta = tower[i - 1]
                g_in = gn_groups(cin, groups)
                gn.in_rows, gn.in_gamma, gn.in_beta = rows_all[i - 1].data_ptr(), _rn.f32(gamma), _rn.f32(beta)
                gn.in_groups, gn.in_act, gn.in_eps = g_in, _rn.ACT[act], eps
            rows = None
            if i < nfold:                               # its output feeds a folded GroupNorm: emit the statistics rows
                g_out = gn_groups(cout, groups)
                rows = torch.empty((nrows, g_out, 4), dtype=torch.float32, device=dev)
                gn.out_rows, gn.out_groups = rows.data_ptr(), g_out
            pre = WINO_PRE.get(w.data_ptr()) if WINO_PRE else None
            if pre is not None:                         # U / Urot transformed ahead of the layer: no kernel-transform blocks here
                torch.cuda.current_stream().wait_event(pre[2])
                v_buf, _ = _winograd_keep_buffers(segs, n, w, training and WINOGRAD_WGRAD, False)
                u_buf = pre[1]
                gn.u_ready, gn.urot_ready = pre[0].data_ptr(), 1
            else:
                v_buf, u_buf = _winograd_keep_buffers(segs, n, w, training and WINOGRAD_WGRAD, training)
            need = L.rn_conv3x3_winograd_workspace(segs, n, cin, cout, tile)
            ws = _rn.workspace(need, dev)
            _rn.check(L.rn_conv3x3_winograd_gn(segs, n, cin, cout, _rn.f32(w), _rn.f32(b) if b is not None else None, tile,
                                               C.byref(gn), ws.data_ptr(), ws.numel(),
                                               _rn.f32(v_buf) if v_buf is not None else None,
                                               _rn.f32(u_buf) if u_buf is not None else None, _rn.stream()),
                      "rn_conv3x3_winograd_gn")
            ys_all.append(ys)
            rows_all.append(rows)
            keep.append((v_buf, u_buf))
            cur = ys
        ctx.cfg, ctx.n = cfg, n
        ctx.keep, ctx.rows, ctx.nrows = keep, rows_all, nrows
        ctx.out_b = out_b
        flat = [t for ys in ys_all[:-1] for t in ys]    # raw outputs of the convs whose GroupNorm is folded
        ctx.save_for_backward(*xs, *flat, *params)
        return tuple(ys_all[-1])

    @staticmethod
    def backward(ctx, *dys):
        groups, eps, act, k, with_out = ctx.cfg
        n = ctx.n
        saved = ctx.saved_tensors
        nconv = k + 1 if with_out else k
        xs = list(saved[:n])
        raw = [list(saved[n * (1 + i):n * (2 + i)]) for i in range(nconv - 1)]
        params = list(saved[n * nconv:])
        tower = [(params[3 * i], params[3 * i + 1], params[3 * i + 2]) for i in range(k)]
        convs = [t[0] for t in tower] + ([params[3 * k]] if with_out else [])
        out_b = params[3 * k + 1] if with_out else None
        L = _rn.lib()
        dev = xs[0].device
        tile = WINOGRAD_TILE
        nfold = k if with_out else k - 1
        inputs = [xs] + raw                              # input tensors of conv i (raw[i-1] = raw output of conv i-1)
        cur_dy = [dy.contiguous() if dy is not None else None for dy in dys]
        for j in range(n):
            if cur_dy[j] is None:
                x = inputs[-1][j]
                cur_dy[j] = torch.zeros((x.shape[0], x.shape[1], x.shape[2], convs[-1].shape[3]), dtype=torch.float32, device=dev)
        grads = [None] * len(params)
        g_rows_group_next = None                         # rows of the GroupNorm after conv i (written by conv i+1's pass)
        for i in range(nconv - 1, -1, -1):
            w = convs[i]
            cin, cout = w.shape[2], w.shape[3]
            x_in = inputs[i]
            dxs = [torch.empty_like(x) for x in x_in]
            segs = _conv_segs(x_in, w, None, None, cur_dy, dxs)
            gnb = _rn.WinoGnBwd()
            grg = grc = None
            if i > 0:                                    # input side folded: dx leaves as g of GroupNorm i-1
                _, gamma, beta = tower[i - 1]
                g_in = gn_groups(cin, groups)
                grg = torch.empty((ctx.nrows, g_in, 2), dtype=torch.float32, device=dev)
                grc = torch.empty((2, ctx.nrows, cin), dtype=torch.float32, device=dev)
                gnb.in_rows, gnb.in_gamma, gnb.in_beta = ctx.rows[i - 1].data_ptr(), _rn.f32(gamma), _rn.f32(beta)
                gnb.in_groups, gnb.in_act, gnb.in_eps = g_in, _rn.ACT[act], eps
                gnb.in_g_rows_group, gnb.in_g_rows_chan = grg.data_ptr(), grc.data_ptr()
            if i < nfold:                                # output side folded: cur_dy holds g of GroupNorm i
                _, gamma_o, _b = tower[i]
                for j in range(n):
                    segs[j].y = _rn.f32(inputs[i + 1][j])
                gnb.out_rows, gnb.out_g_rows_group = ctx.rows[i].data_ptr(), g_rows_group_next.data_ptr()
                gnb.out_gamma, gnb.out_groups, gnb.out_eps = _rn.f32(gamma_o), gn_groups(cout, groups), eps
            v_buf, u_buf = ctx.keep[i]
            dw_buf, dw = _grad_slot(w)
            need = L.rn_conv3x3_winograd_bwd_workspace(segs, n, cin, cout, tile, 1 if v_buf is not None else 0,
                                                       1 if u_buf is not None else 0)
            # the weight-gradient half of the layer (product over the tiles + back-transform: nobody needs dw before the
            # optimizer) can be left for later -- train.Trainer runs it on a side stream beside the backbone's backward pass,
            # whose latency-bound kernels leave most of the chip idle.  It then works from THIS call's workspace: a private one.
            defer = WGRAD_DEFER is not None and dw is None
            ws = torch.empty(int(need), dtype=torch.uint8, device=dev) if defer else _rn.workspace(need, dev)
            gnb.defer_wgrad = 1 if defer else 0
            _rn.check(L.rn_conv3x3_winograd_gn_bwd(segs, n, cin, cout, _rn.f32(w), _rn.f32(dw_buf), 0, tile, C.byref(gnb),
                                                   ws.data_ptr(), ws.numel(),
                                                   _rn.f32(v_buf) if v_buf is not None else None,
                                                   _rn.f32(u_buf) if u_buf is not None else None, _rn.stream()),
                      "rn_conv3x3_winograd_gn_bwd")
            if defer:
                WGRAD_DEFER.append((segs, n, cin, cout, dw_buf, tile, ws, v_buf, 1 if u_buf is not None else 0, x_in))
            grads[3 * i if i < k else 3 * k] = dw
            if i == nconv - 1 and out_b is not None and ctx.needs_input_grad[2 + n + 3 * k + 1]:
                db_buf, db = _grad_slot(out_b)
                geom = _rn.ConvGeom(3, 3, 1, cin, 1)
                need = L.rn_conv2d_bias_grad_workspace(cout)
                wsb = _grad_workspace(need, dev)
                _rn.check(L.rn_conv2d_bias_grad(segs, n, C.byref(geom), _rn.f32(db_buf), wsb.data_ptr(), wsb.numel(), _rn.stream(), _defer_arg()),
                          "rn_conv2d_bias_grad")
                grads[3 * k + 1] = db
            if i > 0:                                    # dbeta / dgamma of GroupNorm i-1 from the per-channel rows
                _, gamma, beta = tower[i - 1]
                dg_buf, dg = _grad_slot(gamma)
                db_buf, db = _grad_slot(beta)
                if _deferring:
                    _deferred_keep.append(grc)
                _rn.check(L.rn_reduce_rows(_rn.f32(grc[0]), _rn.f32(db_buf), cin, ctx.nrows, 0, _rn.stream(), _defer_arg()), "rn_reduce_rows")
                _rn.check(L.rn_reduce_rows(_rn.f32(grc[1]), _rn.f32(dg_buf), cin, ctx.nrows, 0, _rn.stream(), _defer_arg()), "rn_reduce_rows")
                grads[3 * (i - 1) + 1], grads[3 * (i - 1) + 2] = dg, db
            cur_dy = dxs
            g_rows_group_next = grg
        return (None, None) + tuple(cur_dy) + tuple(grads)


# None: every Winograd-tower layer finishes its weight gradient inside its backward call.  A list: the calls record their
# weight-gradient halves here instead (see _WinoTower.backward) and run_deferred_wgrads launches them.
WGRAD_DEFER = None


def run_deferred_wgrads(records):
    """Launch the recorded weight-gradient halves (rn_conv3x3_winograd_gn_bwd_wgrad) on the CURRENT stream, in order."""
    L = _rn.lib()
    for segs, n, cin, cout, dw_buf, tile, ws, v_buf, urot_given, _keep in records:
        _rn.check(L.rn_conv3x3_winograd_gn_bwd_wgrad(segs, n, cin, cout, _rn.f32(dw_buf), 0, tile, ws.data_ptr(), ws.numel(),
                                                     _rn.f32(v_buf) if v_buf is not None else None, urot_given, _rn.stream()),
                  "rn_conv3x3_winograd_gn_bwd_wgrad")


def wino_tower(xs, tower, out_w=None, out_b=None, groups=32, eps=1e-5, act=None):
    """[conv3x3 -> GroupNorm -> act] for every (w, gamma, beta) of `tower`, then (if out_w is given) conv3x3(out_w) + out_b,
    applied to every tensor of the list `xs` (shared parameters), as GroupNorm-folded Winograd layers.  Without out_w the
    result is the RAW output of the last conv: its GroupNorm + activation are left to the caller."""
    k = len(tower)
    cfg = (groups, float(eps), act, k, out_w is not None)
    flat = [t for layer in tower for t in layer]
    if out_w is not None:
        flat += [out_w, out_b]
    return list(_WinoTower.apply(cfg, len(xs), *xs, *flat))


class _Depthwise(torch.autograd.Function):
    """tf.nn.depthwise_conv2d(padding='SAME'), kernel [k,k,C,1] (mobilenet_v2.py:35-36)."""

    @staticmethod
    def forward(ctx, x, w, stride, gn=None):
        k = w.shape[0]
        n, h, wd, c = x.shape
        assert w.shape[2] == c and w.shape[3] == 1
        x = x.contiguous()
        oh, _ = _rn.same_pad(h, k, stride)
        ow, _ = _rn.same_pad(wd, k, stride)
        y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
        L = _rn.lib()
        ctx.stride = stride
        ctx.save_for_backward(x, w)
        ctx.set_materialize_grads(False)
        if gn is not None:
            g = gn_groups(c, gn[0])
            lay = _rn.GnRows(None, 0, 0, g)
            nbytes = L.rn_depthwise_stats_rows(n, h, wd, c, k, stride, g, C.byref(lay))
            if nbytes:
                rows = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
                lay.rows = rows.data_ptr()
                _rn.check(L.rn_depthwise_fwd_stats(_rn.f32(x), _rn.f32(w), _rn.f32(y), n, h, wd, c, k, stride, C.byref(lay),
                                                   _rn.stream()), "rn_depthwise_fwd_stats")
                _LAST_ROWS_LAYOUT[0] = (lay.rows_per_sample, lay.per_group, g)
                ctx.mark_non_differentiable(rows)
                return y, rows
        _rn.check(L.rn_depthwise_fwd(_rn.f32(x), _rn.f32(w), _rn.f32(y), n, h, wd, c, k, stride, _rn.stream()), "rn_depthwise_fwd")
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:
            return None, None, None, None
        x, w = ctx.saved_tensors
        k = w.shape[0]
        n, h, wd, c = x.shape
        dy = dy.contiguous()
        L = _rn.lib()
        dx = dw = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and k == 3 and not WGRAD_SIDE_STREAM:
            dx = torch.empty_like(x)                                  # both gradients from one launch
            dw_buf, dw = _grad_slot(w)
            need = L.rn_depthwise_wgrad_workspace(n, h, wd, c, k, ctx.stride)
            ws = _grad_workspace(need, x.device)
            _rn.check(L.rn_depthwise_bwd(_rn.f32(x), _rn.f32(dy), _rn.f32(w), _rn.f32(dx), _rn.f32(dw_buf), n, h, wd, c, k,
                                         ctx.stride, ws.data_ptr(), ws.numel(), _rn.stream(), _defer_arg()), "rn_depthwise_bwd")
            return dx, dw, None, None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _rn.check(L.rn_depthwise_dgrad(_rn.f32(dy), _rn.f32(w), _rn.f32(dx), n, h, wd, c, k, ctx.stride,
                                           _rn.stream()), "rn_depthwise_dgrad")
        if ctx.needs_input_grad[1]:
            dw_buf, dw = _grad_slot(w)
            with _on_side_stream(x.device, [x, dy, dw_buf], direct=(dw is None)):
                need = L.rn_depthwise_wgrad_workspace(n, h, wd, c, k, ctx.stride)
                ws = _grad_workspace(need, x.device)
                _rn.check(L.rn_depthwise_wgrad(_rn.f32(x), _rn.f32(dy), _rn.f32(dw_buf), n, h, wd, c, k, ctx.stride,
                                               ws.data_ptr(), ws.numel(), _rn.stream(), _defer_arg()), "rn_depthwise_wgrad")
        return dx, dw, None, None


def depthwise_conv2d(x, w, stride=1, gn=None):
    """gn = (groups, eps) of the GroupNorm that follows: its statistics come out of the same kernel (see conv2d)."""
    if gn is not None and GN_PRODUCER_STATS and x.is_cuda and x.dtype == torch.float32:
        outs = _Depthwise.apply(x, w, stride, (int(gn[0]), float(gn[1])))
        if isinstance(outs, tuple):
            y = outs[0]
            y._gn_rows = (outs[1],) + _LAST_ROWS_LAYOUT[0]
            return y
        return outs
    return _Depthwise.apply(x, w, stride)


# GroupNorm -> depthwise 3x3 -> GroupNorm of a MobileNetV2 bottleneck as one kernel per direction (rn_dwgn_fwd / rn_dwgn_bwd).
# Correct (tests/test_gpu_ops.py::test_fused_groupnorm_depthwise_groupnorm) but OFF by default: measured on the headline step it
# is slower than the three kernels it replaces (347 vs 360 images/s; forward 19-40 us vs ~27, backward 58 vs ~41 per
# bottleneck): one block per (sample, group) is 64 blocks on 256 CUs, and with one channel lane per thread the stencil's index
# arithmetic (~100 VALU instructions per output) makes those 64 CUs ALU-bound.  See DESIGN.md section 9.
DW_GN_FUSED = os.environ.get("RN_DW_GN_FUSED", "0") == "1"


def _dwgn_params(x, stride, groups, eps, act, rate, seed1, seed2, seed_dev):
    n, h, w, c = x if isinstance(x, tuple) else x.shape
    return _rn.DwGnParams(n, h, w, c, stride, groups, _rn.ACT[act], eps, rate, seed1, seed2,
                          seed_dev.data_ptr() if seed_dev is not None else None)


def dw_gn_ok(shape, w, stride, groups, act, need_backward):
    """Can drop(act(GN(dw3x3(drop(act(GN(x))))))) run as the fused kernels for an fp32 device tensor x of `shape` [n,h,w,c]?"""
    if not (DW_GN_FUSED and w.is_cuda and len(shape) == 4 and w.shape[0] == 3 and w.shape[1] == 3 and w.shape[2] == shape[3]
            and stride in (1, 2) and act in (None, "none", "relu", "elu", "relu6")):
        return False
    p = _dwgn_params(tuple(int(v) for v in shape), stride, groups, 1e-5, act, 0.0, 0, 0, None)
    return bool(_rn.lib().rn_dwgn_supported(C.byref(p), 1 if need_backward else 0))


class _DwGnFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, x, g1, b1, w, g2, b2):
        stride, groups, eps, act, rate, seed1, seed2, seed_dev = cfg
        x = x.contiguous()
        n, h, wd, c = x.shape
        oh, _ = _rn.same_pad(h, 3, stride)
        ow, _ = _rn.same_pad(wd, 3, stride)
        y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
        stats = torch.empty((4, n, groups), dtype=torch.float32, device=x.device)
        p = _dwgn_params(x, stride, groups, eps, act, rate, seed1, seed2, seed_dev)
        _rn.check(_rn.lib().rn_dwgn_fwd(_rn.f32(x), _rn.f32(g1), _rn.f32(b1), _rn.f32(w), _rn.f32(g2), _rn.f32(b2), _rn.f32(y),
                                        _rn.f32(stats), C.byref(p), _rn.stream()), "rn_dwgn_fwd")
        ctx.cfg = cfg
        ctx.save_for_backward(x, g1, b1, w, g2, b2, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, groups, eps, act, rate, seed1, seed2, seed_dev = ctx.cfg
        x, g1, b1, w, g2, b2, stats = ctx.saved_tensors
        n, h, wd, c = x.shape
        L = _rn.lib()
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        rows = torch.empty((13 * n * c,), dtype=torch.float32, device=x.device)
        if _deferring:
            _deferred_keep.append(rows)
        p = _dwgn_params(x, stride, groups, eps, act, rate, seed1, seed2, seed_dev)
        _rn.check(L.rn_dwgn_bwd(_rn.f32(x), _rn.f32(dy), _rn.f32(g1), _rn.f32(b1), _rn.f32(w), _rn.f32(g2), _rn.f32(b2),
                                _rn.f32(stats), _rn.f32(dx), _rn.f32(rows), C.byref(p), _rn.stream()), "rn_dwgn_bwd")
        grads = []
        nc = n * c
        for i, prm in enumerate((g1, b1, g2, b2, w)):
            buf, ret = _grad_slot(prm)
            count = 9 * c if i == 4 else c
            sec = rows[i * nc:i * nc + n * count]
            _rn.check(L.rn_reduce_rows(_rn.f32(sec), _rn.f32(buf), count, n, 0, _rn.stream(), _defer_arg()), "rn_reduce_rows")
            grads.append(ret)
        dg1, db1, dg2, db2, dw = grads                    # rows order; the inputs are (x, g1, b1, w, g2, b2)
        return None, dx, dg1, db1, dw, dg2, db2


def dw_gn_fused(x, gamma1, beta1, w, gamma2, beta2, stride, groups, eps, act, rate=0.0, seed1=0, seed2=0, seed_dev=None):
    """drop2(act(GN2(depthwise3x3(drop1(act(GN1(x))))))): see rn_dwgn_fwd."""
    cfg = (int(stride), int(groups), float(eps), act, float(rate), int(seed1), int(seed2), seed_dev)
    return _DwGnFused.apply(cfg, x, gamma1, beta1, w, gamma2, beta2)


def gn_groups(c, groups=32):
    """Largest divisor of c that is <= min(groups, c): the reference's min(32, C) rule
    (normalization.py:24) made total for C=144, where the reference itself cannot run
    (SURVEY Q2)."""
    g = min(groups, c)
    while c % g:
        g -= 1
    return g


# single-kernel GroupNorm for mid-sized maps (<= 128 co-resident blocks exchanging tagged sums, bounded polls).  OFF by
# default: its blocks wait for each other, which is only safe while nothing else holds CUs (a collective under the backward
# pass does) -- the default configuration contains no kernel that waits for another block.  RN_GN_GRID_RESIDENT=1 opts in.
GN_GRID_RESIDENT = os.environ.get("RN_GN_GRID_RESIDENT", "0") == "1"


def _gn_params(c, g, eps, act, drop_rate, seed, seed_dev, act_after_residual, device=None):
    sync = _rn.sync_counters(device).data_ptr() if (GN_GRID_RESIDENT and device is not None and device.type == 'cuda') else None
    return _rn.GnParams(c=c, groups=g, act=_rn.ACT[act], act_after_residual=1 if act_after_residual else 0, eps=eps,
                        drop_rate=drop_rate, drop_seed=seed,
                        drop_seed_dev=seed_dev.data_ptr() if seed_dev is not None else None, sync=sync)


def _gn_segs(xs, ys, ress, dys, dxs, means, rstds, dress=None):
    segs = (_rn.GnSeg * len(xs))()
    for i, x in enumerate(xs):
        s = segs[i]
        s.dresidual = _rn.f32(dress[i]) if dress is not None and dress[i] is not None else None
        s.x = _rn.f32(x)
        s.y = _rn.f32(ys[i]) if ys is not None else None
        s.residual = _rn.f32(ress[i]) if ress is not None and ress[i] is not None else None
        s.dy = _rn.f32(dys[i]) if dys is not None else None
        s.dx = _rn.f32(dxs[i]) if dxs is not None else None
        s.mean = _rn.f32(means[i])
        s.rstd = _rn.f32(rstds[i])
        s.n = x.shape[0]
        s.hw = x.shape[1] * x.shape[2]
    return segs


class _GroupNormAct(torch.autograd.Function):
    """y_i = dropout(act(GN(x_i))) + residual_i for n tensors sharing gamma/beta."""

    @staticmethod
    def forward(ctx, cfg, gamma, beta, n, *tensors):
        groups, eps, act, drop_rate, seed, seed_dev, aar = cfg[:7]
        rows_layout = cfg[7] if len(cfg) > 7 else None   # tensors end with the producer's partial-sum rows of the single x
        have_stats = rows_layout is not None
        xs = [t.contiguous() for t in tensors[:n]]
        ress = [t.contiguous() if t is not None else None for t in tensors[n:2 * n]]
        c = xs[0].shape[3]
        g = gn_groups(c, groups)
        L = _rn.lib()
        dev = xs[0].device
        ys = [torch.empty_like(x) for x in xs]
        means = [torch.empty((x.shape[0], g), dtype=torch.float32, device=dev) for x in xs]
        rstds = [torch.empty((x.shape[0], g), dtype=torch.float32, device=dev) for x in xs]
        for x, r in zip(xs, ress):
            assert x.shape[3] == c and (r is None or r.shape == x.shape)
        params = _gn_params(c, g, eps, act, drop_rate, seed, seed_dev, aar, dev)
        if have_stats:
            lay = _rn.GnRows(tensors[2 * n].data_ptr(), rows_layout[0], rows_layout[1], rows_layout[2])
            params.stat_rows = C.addressof(lay)
        ctx.extra = 1 if have_stats else 0
        segs = _gn_segs(xs, ys, ress, None, None, means, rstds)
        need = L.rn_group_norm_workspace(segs, n, C.byref(params))
        ws = _rn.workspace(need, dev)
        _rn.check(L.rn_group_norm_fwd(segs, n, C.byref(params), _rn.f32(gamma), _rn.f32(beta), ws.data_ptr(),
                                      ws.numel(), _rn.stream()), "rn_group_norm_fwd")
        ctx.cfg = (c, g, eps, act, drop_rate, seed, seed_dev, aar)
        ctx.n = n
        ctx.has_res = [r is not None for r in ress]
        # act-after-residual needs the residual again in backward (act'(GN(x) + r))
        ctx.res_saved = [r if (aar and r is not None) else None for r in ress]
        ctx.save_for_backward(gamma, beta, *xs, *means, *rstds)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        saved = ctx.saved_tensors
        gamma, beta = saved[0], saved[1]
        xs = list(saved[2:2 + n])
        means = list(saved[2 + n:2 + 2 * n])
        rstds = list(saved[2 + 2 * n:2 + 3 * n])
        c, g, eps, act, drop_rate, seed, seed_dev, aar = ctx.cfg
        L = _rn.lib()
        dev = xs[0].device
        dys = [dy.contiguous() if dy is not None else torch.zeros_like(x) for dy, x in zip(dys, xs)]
        dxs = [torch.empty_like(x) for x in xs]
        ress = ctx.res_saved
        dress = [torch.empty_like(x) if r is not None else None for x, r in zip(xs, ress)]
        dgamma_buf, dgamma = _grad_slot(gamma)
        dbeta_buf, dbeta = _grad_slot(beta)
        params = _gn_params(c, g, eps, act, drop_rate, seed, seed_dev, aar, dev)
        segs = _gn_segs(xs, None, ress, dys, dxs, means, rstds, dress)
        need = L.rn_group_norm_workspace(segs, n, C.byref(params))
        ws = _grad_workspace(need, dev)
        _rn.check(L.rn_group_norm_bwd(segs, n, C.byref(params), _rn.f32(gamma), _rn.f32(beta), _rn.f32(dgamma_buf),
                                      _rn.f32(dbeta_buf), ws.data_ptr(), ws.numel(), _rn.stream(), _defer_arg()),
                  "rn_group_norm_bwd")
        dres = [(dress[i] if dress[i] is not None else dys[i]) if ctx.has_res[i] else None for i in range(n)]
        return (None, dgamma, dbeta, None) + tuple(dxs) + tuple(dres) + (None,) * ctx.extra


def group_norm_act(x, gamma, beta, groups=32, eps=1e-5, act=None, residual=None, drop_rate=0.0, seed=0,
                   seed_dev=None, act_after_residual=False):
    """Fused GroupNorm -> activation -> dropout -> (+ residual); with act_after_residual the order is
    GroupNorm -> (+ residual) -> activation (ResNeXt).  `x` / `residual` may be lists."""
    multi = isinstance(x, (list, tuple))
    xs = _as_list(x)
    ress = _as_list(residual) if residual is not None else [None] * len(xs)
    cfg = (groups, float(eps), act, float(drop_rate), int(seed), seed_dev, bool(act_after_residual))
    rows = getattr(xs[0], '_gn_rows', None) if len(xs) == 1 else None
    if rows is not None and rows[3] == gn_groups(xs[0].shape[3], groups) and xs[0].is_contiguous():
        # the conv / depthwise kernel that produced x also wrote its partial sums: merge + apply in one pass over x
        ys = _GroupNormAct.apply(cfg + (tuple(rows[1:]),), gamma, beta, 1, xs[0], ress[0], rows[0])
    else:
        ys = _GroupNormAct.apply(cfg, gamma, beta, len(xs), *xs, *ress)
    return list(ys) if multi else ys[0]


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous()
        y = torch.empty_like(x)
        _rn.check(_rn.lib().rn_act_fwd(_rn.f32(x), _rn.f32(y), x.numel(), _rn.ACT[act], _rn.stream()), "rn_act_fwd")
        ctx.act = act
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _rn.check(_rn.lib().rn_act_bwd(_rn.f32(x), _rn.f32(dy), _rn.f32(dx), x.numel(), _rn.ACT[ctx.act],
                                       _rn.stream()), "rn_act_bwd")
        return dx, None


def activation(x, act):
    if _rn.ACT[act] == 0:
        return x
    return _Act.apply(x, act)


class _UpsampleAdd(torch.autograd.Function):
    """lateral + resize_nearest(top -> lateral size, align_corners=True) (retinanet.py:153-157)."""

    @staticmethod
    def forward(ctx, lateral, top):
        lateral = lateral.contiguous()
        top = top.contiguous()
        n, h, w, c = lateral.shape
        assert top.shape[0] == n and top.shape[3] == c
        y = torch.empty_like(lateral)
        _rn.check(_rn.lib().rn_upsample_add_fwd(_rn.f32(lateral), _rn.f32(top), _rn.f32(y), n, h, w, top.shape[1],
                                                top.shape[2], c, _rn.stream()), "rn_upsample_add_fwd")
        ctx.shapes = (n, h, w, top.shape[1], top.shape[2], c)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, th, tw, c = ctx.shapes
        dy = dy.contiguous()
        dtop = None
        if ctx.needs_input_grad[1]:
            dtop = torch.empty((n, th, tw, c), dtype=torch.float32, device=dy.device)
            _rn.check(_rn.lib().rn_upsample_add_bwd_top(_rn.f32(dy), _rn.f32(dtop), n, h, w, th, tw, c, _rn.stream()),
                      "rn_upsample_add_bwd_top")
        return (dy if ctx.needs_input_grad[0] else None), dtop


def upsample_add(lateral, top):
    if lateral.dtype == torch.float16:
        import ops_f16
        return ops_f16.upsample_add(lateral, top)
    return _UpsampleAdd.apply(lateral, top)


def _loss_segs(cls_logits, cls_labels, reg_preds, reg_labels, masks, dcls, dreg):
    n = len(cls_logits)
    segs = (_rn.LossSeg * n)()
    for i in range(n):
        s = segs[i]
        s.cls_logit = _rn.f32(cls_logits[i])
        s.cls_label = _rn.f32(cls_labels[i])
        s.reg_pred = _rn.f32(reg_preds[i])
        s.reg_label = _rn.f32(reg_labels[i])
        s.trainable = _rn.ptr(masks[i])
        s.d_cls_logit = _rn.f32(dcls[i]) if dcls is not None else None
        s.d_reg_pred = _rn.f32(dreg[i]) if dreg is not None else None
        s.rows = masks[i].numel()
    return segs


class _DetectionLoss(torch.autograd.Function):
    """(class_loss, regr_loss, stats) over all levels; labels and masks carry no gradient."""

    @staticmethod
    def forward(ctx, mode, num_classes, n, *tensors):
        cls_logits = [t.contiguous() for t in tensors[:n]]
        reg_preds = [t.contiguous() for t in tensors[n:2 * n]]
        cls_labels = [t.contiguous() for t in tensors[2 * n:3 * n]]
        reg_labels = [t.contiguous() for t in tensors[3 * n:4 * n]]
        masks = [t.contiguous() for t in tensors[4 * n:5 * n]]
        for m in masks:
            assert m.dtype in (torch.uint8, torch.bool)
        for z, l, m in zip(cls_logits, cls_labels, masks):
            assert z.shape == l.shape and z.shape[-1] == num_classes and z.numel() == m.numel() * num_classes
        L = _rn.lib()
        dev = cls_logits[0].device
        stats = torch.empty((_rn.LOSS_STATS_HEADER + 3 * num_classes,), dtype=torch.float32, device=dev)
        segs = _loss_segs(cls_logits, cls_labels, reg_preds, reg_labels, masks, None, None)
        need = L.rn_loss_workspace(segs, n, num_classes)
        ws = _rn.workspace(need, dev)
        class_loss = torch.empty((), dtype=torch.float32, device=dev)     # stand-alone scalars written by the kernel itself
        regr_loss = torch.empty((), dtype=torch.float32, device=dev)      # (no copies of stats[0], stats[1])
        _rn.check(L.rn_loss_fwd(segs, n, num_classes, _rn.LOSS_MODE[mode], _rn.f32(stats), _rn.f32(class_loss), _rn.f32(regr_loss),
                                ws.data_ptr(), ws.numel(), _rn.stream()), "rn_loss_fwd")
        ctx.mode, ctx.num_classes, ctx.n = mode, num_classes, n
        ctx.save_for_backward(stats, *cls_logits, *reg_preds, *cls_labels, *reg_labels, *masks)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)       # no zero tensor for the non-differentiable `stats` output
        return class_loss, regr_loss, stats

    @staticmethod
    def backward(ctx, g_cls, g_reg, _g_stats):
        n = ctx.n
        saved = ctx.saved_tensors
        stats = saved[0]
        cls_logits = list(saved[1:1 + n])
        reg_preds = list(saved[1 + n:1 + 2 * n])
        cls_labels = list(saved[1 + 2 * n:1 + 3 * n])
        reg_labels = list(saved[1 + 3 * n:1 + 4 * n])
        masks = list(saved[1 + 4 * n:1 + 5 * n])
        dev = stats.device
        zero = None
        if g_cls is None or g_reg is None:
            zero = torch.zeros((1,), dtype=torch.float32, device=dev)
        g_cls = g_cls.reshape(1).contiguous().float() if g_cls is not None else zero
        g_reg = g_reg.reshape(1).contiguous().float() if g_reg is not None else zero
        dcls = [torch.empty_like(z) for z in cls_logits]
        dreg = [torch.empty_like(r) for r in reg_preds]
        segs = _loss_segs(cls_logits, cls_labels, reg_preds, reg_labels, masks, dcls, dreg)
        _rn.check(_rn.lib().rn_loss_bwd(segs, n, ctx.num_classes, _rn.LOSS_MODE[ctx.mode], _rn.f32(stats),
                                        _rn.f32(g_cls), _rn.f32(g_reg), _rn.stream()), "rn_loss_bwd")
        return (None, None, None) + tuple(dcls) + tuple(dreg) + (None,) * (3 * n)


def detection_loss(cls_logits, reg_preds, cls_labels, reg_labels, trainable_masks, num_classes, mode="bce_dice"):
    """Lists over pyramid levels (P3..P7 order).  Returns (class_loss, regr_loss, stats)."""
    n = len(cls_logits)
    return _DetectionLoss.apply(mode, num_classes, n, *cls_logits, *reg_preds, *cls_labels, *reg_labels,
                                *trainable_masks)


class _Dropout(torch.autograd.Function):
    """tf.layers.Dropout on its own (DenseNet: after a conv).  Same counter-based mask both ways."""

    @staticmethod
    def forward(ctx, x, rate, seed, seed_dev):
        x = x.contiguous()
        y = torch.empty_like(x)
        _rn.check(_rn.lib().rn_dropout(_rn.f32(x), _rn.f32(y), x.numel(), rate, seed,
                                       seed_dev.data_ptr() if seed_dev is not None else None, _rn.stream()), "rn_dropout")
        ctx.cfg = (rate, seed, seed_dev)
        return y

    @staticmethod
    def backward(ctx, dy):
        rate, seed, seed_dev = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _rn.check(_rn.lib().rn_dropout(_rn.f32(dy), _rn.f32(dx), dy.numel(), rate, seed,
                                       seed_dev.data_ptr() if seed_dev is not None else None, _rn.stream()), "rn_dropout")
        return dx, None, None, None


DROPOUT_COUNTER_STEP = 0x9E3779B9


def advance_dropout_counter(counter):
    """Next step's dropout masks: bump the device-side word every fused dropout hashes with.  train.Trainer.step does not
    need this (its optimizer kernel bumps the counter, rn_optimizer_step); it is for backward passes without an update."""
    _rn.check(_rn.lib().rn_counter_add(counter.data_ptr(), DROPOUT_COUNTER_STEP, _rn.stream()), "rn_counter_add")


class _Fanout(torch.autograd.Function):
    """k views of the same tensors for k consumers; backward sums the k gradients with ONE launch of our add kernel
    (rn_add_segs) instead of leaving the sums to autograd's accumulation (PyTorch kernels inside the step)."""

    @staticmethod
    def forward(ctx, k, n, *xs):
        ctx.k, ctx.n = k, n
        ctx.set_materialize_grads(False)       # a branch without a gradient stays None (no zero tensors to add)
        return tuple(x.view_as(x) for _ in range(k) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        k, n = ctx.k, ctx.n
        acc = [gs[j] for j in range(n)]
        for i in range(1, k):
            nxt = [gs[i * n + j] for j in range(n)]
            pairs = [(j, a, b) for j, (a, b) in enumerate(zip(acc, nxt)) if a is not None and b is not None]
            for j, (a, b) in enumerate(zip(acc, nxt)):
                if a is None:
                    acc[j] = b
            for c0 in range(0, len(pairs), _rn.MAX_SEG):
                chunk = pairs[c0:c0 + _rn.MAX_SEG]
                segs = (_rn.AddSeg * len(chunk))()
                outs = []
                for q, (j, a, b) in enumerate(chunk):
                    a, b = a.contiguous(), b.contiguous()
                    o = torch.empty_like(a)
                    segs[q] = _rn.AddSeg(_rn.f32(a), _rn.f32(b), _rn.f32(o), a.numel())
                    outs.append((j, o, a, b))
                _rn.check(_rn.lib().rn_add_segs(segs, len(chunk), _rn.stream()), "rn_add_segs")
                for j, o, _a, _b in outs:
                    acc[j] = o
        return (None, None) + tuple(acc)


class _Add(torch.autograd.Function):
    """a + b (lists: one launch for all pairs) by rn_add_segs; the gradient passes through to both."""

    @staticmethod
    def forward(ctx, n, *ts):
        ctx.set_materialize_grads(False)
        outs = []
        for c0 in range(0, n, _rn.MAX_SEG):
            m = min(_rn.MAX_SEG, n - c0)
            segs = (_rn.AddSeg * m)()
            for q in range(m):
                a, b = ts[c0 + q].contiguous(), ts[n + c0 + q].contiguous()
                assert a.shape == b.shape
                o = torch.empty_like(a)
                segs[q] = _rn.AddSeg(_rn.f32(a), _rn.f32(b), _rn.f32(o), a.numel())
                outs.append(o)
            _rn.check(_rn.lib().rn_add_segs(segs, m, _rn.stream()), "rn_add_segs")
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        return (None,) + tuple(gs) + tuple(gs)


def add(a, b):
    if isinstance(a, (list, tuple)):
        return list(_Add.apply(len(a), *a, *b))
    return _Add.apply(1, a, b)[0]


def fanout(xs, k=2):
    """`xs` (tensor or list) for k consumers: returns k copies of the structure (views).  Gradients are summed by rn_add_segs."""
    multi = isinstance(xs, (list, tuple))
    lst = list(xs) if multi else [xs]
    if not any(x.requires_grad for x in lst) or not torch.is_grad_enabled():
        return [lst if multi else lst[0] for _ in range(k)]
    out = _Fanout.apply(k, len(lst), *lst)
    n = len(lst)
    groups = [list(out[i * n:(i + 1) * n]) for i in range(k)]
    return [g if multi else g[0] for g in groups]


def dropout(x, rate, seed=0, seed_dev=None):
    if rate == 0.0:
        return x
    return _Dropout.apply(x, float(rate), int(seed), seed_dev)


# ---------------------------------------------------------------------------------------- concat-free DenseNet block
def _gn_raw(fwd, x, x_ld, c, n, hw, gamma, beta, groups, eps, act, mean, rstd, y=None, dy=None, dx=None, dx_ld=0, dx_acc=False,
            dgamma=None, dbeta=None, stat_rows=None):
    """One rn_group_norm_fwd / _bwd call on raw buffers (x may be the first c channels of a buffer with x_ld channels).
    stat_rows (forward, dense x): the _rn.GnRows its producer wrote (rn_conv2d_fwd_stats / rn_conv2d_fwd_dropout)."""
    L = _rn.lib()
    dev = gamma.device
    g = gn_groups(c, groups)
    params = _gn_params(c, g, eps, act, 0.0, 0, None, False, dev)
    if stat_rows is not None and fwd:
        params.stat_rows = C.addressof(stat_rows)
    segs = (_rn.GnSeg * 1)()
    sg = segs[0]
    sg.x, sg.mean, sg.rstd, sg.n, sg.hw, sg.x_ld = x.data_ptr(), _rn.f32(mean), _rn.f32(rstd), n, hw, x_ld
    if fwd:
        sg.y = _rn.f32(y)
        ws = _rn.workspace(L.rn_group_norm_workspace(segs, 1, C.byref(params)), dev)
        _rn.check(L.rn_group_norm_fwd(segs, 1, C.byref(params), _rn.f32(gamma), _rn.f32(beta), ws.data_ptr(), ws.numel(), _rn.stream()),
                  "rn_group_norm_fwd")
    else:
        sg.dy, sg.dx, sg.dx_ld, sg.dx_accumulate = _rn.f32(dy), dx.data_ptr(), dx_ld, 1 if dx_acc else 0
        ws = _grad_workspace(L.rn_group_norm_workspace(segs, 1, C.byref(params)), dev)
        _rn.check(L.rn_group_norm_bwd(segs, 1, C.byref(params), _rn.f32(gamma), _rn.f32(beta), _rn.f32(dgamma), _rn.f32(dbeta),
                                      ws.data_ptr(), ws.numel(), _rn.stream(), _defer_arg()), "rn_group_norm_bwd")


def _conv_bwd_raw(x, w, dy, stride=1):
    """dx and dw of a dense conv through the merged kernel when it applies (rn_conv2d_bwd), else dgrad + wgrad."""
    L = _rn.lib()
    kh, kw, cin, cout = w.shape
    geom = _rn.ConvGeom(kh, kw, stride, cin, 1)
    dx = torch.empty_like(x)
    segs = _conv_segs([x], w, None, None, [dy], [dx])
    dw_buf, dw = _grad_slot(w)
    need = L.rn_conv2d_wgrad_workspace(segs, 1, C.byref(geom))
    ws = _grad_workspace(need, w.device)
    if MERGED_CONV_BWD and L.rn_conv2d_dgrad_workspace(segs, 1, C.byref(geom)) == 0:
        _rn.check(L.rn_conv2d_bwd(segs, 1, C.byref(geom), _rn.f32(dw_buf), ws.data_ptr(), ws.numel(), _rn.stream(), _defer_arg()),
                  "rn_conv2d_bwd")
    else:
        _conv_dgrad(segs, 1, geom, w.device)
        _rn.check(L.rn_conv2d_wgrad(segs, 1, C.byref(geom), _rn.f32(dw_buf), 0, ws.data_ptr(), ws.numel(), _rn.stream(), _defer_arg()),
                  "rn_conv2d_wgrad")
    return dx, dw


DENSE_FUSED_DROPOUT = os.environ.get("RN_DENSE_FUSED_DROPOUT", "1") == "1"     # (A/B aid: 0 = conv, rn_dropout and a statistics pass as three launches)


class _DenseBlock(torch.autograd.Function):
    """A DenseNet-BC block (densenet.py:83-121) without the growth concat: ONE [n,h,w,c_total] buffer; layer i
    (GN-act-1x1(4k)-drop-GN-act-3x3(k)-drop, densenet.py:50-80) normalises the first c_i channels in place (rn_gn_seg.x_ld) and
    its k output channels are written straight into their slice (rn_dropout_strided).  Backward: one gradient buffer; each
    layer reads its slice and ADDS the gradient of its input prefix (rn_gn_seg.dx_accumulate), last layer first."""

    @staticmethod
    def forward(ctx, cfg, x, *params):
        depth, k, groups, eps, act, rate, seeds, seed_dev = cfg
        L = _rn.lib()
        x = x.contiguous()
        n, h, w, c_in = x.shape
        dev = x.device
        hw, px = h * w, n * h * w
        ct = c_in + depth * k
        sd = seed_dev.data_ptr() if seed_dev is not None else None
        buf = torch.empty((n, h, w, ct), dtype=torch.float32, device=dev)
        _rn.check(L.rn_dropout_strided(_rn.f32(x), _rn.f32(buf), px, c_in, c_in, 0, ct, 0, 0.0, 0, None, _rn.stream()), "rn_dropout_strided")
        saved = []
        for i in range(depth):
            g1, b1, w1, g2, b2, w2 = params[6 * i:6 * i + 6]
            ci, c4 = c_in + i * k, w1.shape[3]
            gr1, gr2 = gn_groups(ci, groups), gn_groups(c4, groups)
            a = torch.empty((n, h, w, ci), dtype=torch.float32, device=dev)
            m1, r1 = torch.empty((n, gr1), device=dev), torch.empty((n, gr1), device=dev)
            _gn_raw(True, buf, ct, ci, n, hw, g1, b1, groups, eps, act, m1, r1, y=a)
            y1 = torch.empty((n, h, w, c4), dtype=torch.float32, device=dev)
            segs1, geom1 = _conv_segs([a], w1, None, [y1], None, None), _rn.ConvGeom(1, 1, 1, ci, 1)
            d1, lay2, rows2 = None, None, None
            if rate > 0.0 and DENSE_FUSED_DROPOUT:
                # conv -> Dropout -> the statistics of GroupNorm 2 in ONE launch (rn_conv2d_fwd_dropout): y1 is never stored un-dropped, the
                # dropout pass (read + write of the 4k-channel tensor) and the GroupNorm's statistics pass (another read) are gone
                ok, lay = C.c_int(0), _rn.GnRows(None, 0, 0, gr2)
                nbytes = L.rn_conv2d_dropout_rows(segs1, 1, C.byref(geom1), gr2, C.byref(lay), C.byref(ok))
                if ok.value:
                    if nbytes:
                        rows2 = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
                        lay.rows, lay2 = rows2.data_ptr(), lay
                    _rn.check(L.rn_conv2d_fwd_dropout(segs1, 1, C.byref(geom1), rate, seeds[i][0], sd, C.byref(lay) if nbytes else None,
                                                      _rn.stream()), "rn_conv2d_fwd_dropout")
                    d1 = y1
            if d1 is None:
                got = _conv_fwd(segs1, 1, geom1, dev, gn=(groups, eps) if rate <= 0.0 else None)
                d1 = y1
                if got is not None:                   # (no dropout: the conv's own epilogue wrote the rows of y1)
                    rows2 = got[0]
                    lay2 = _rn.GnRows(rows2.data_ptr(), got[1], got[2], got[3])
                if rate > 0.0:
                    d1 = torch.empty_like(y1)
                    _rn.check(L.rn_dropout(_rn.f32(y1), _rn.f32(d1), y1.numel(), rate, seeds[i][0], sd, _rn.stream()), "rn_dropout")
            a2 = torch.empty_like(d1)
            m2, r2 = torch.empty((n, gr2), device=dev), torch.empty((n, gr2), device=dev)
            _gn_raw(True, d1, 0, c4, n, hw, g2, b2, groups, eps, act, m2, r2, y=a2, stat_rows=lay2)
            y2 = torch.empty((n, h, w, k), dtype=torch.float32, device=dev)
            _conv_fwd(_conv_segs([a2], w2, None, [y2], None, None), 1, _rn.ConvGeom(3, 3, 1, c4, 1), dev)
            _rn.check(L.rn_dropout_strided(_rn.f32(y2), _rn.f32(buf), px, k, k, 0, ct, ci, rate, seeds[i][1], sd, _rn.stream()),
                      "rn_dropout_strided")
            saved += [a, m1, r1, d1, a2, m2, r2]
        ctx.cfg = cfg
        ctx.nparams = len(params)
        ctx.save_for_backward(buf, *params, *saved)
        return buf

    @staticmethod
    def backward(ctx, dbuf_in):
        depth, k, groups, eps, act, rate, seeds, seed_dev = ctx.cfg
        L = _rn.lib()
        tensors = ctx.saved_tensors
        buf = tensors[0]
        params = tensors[1:1 + ctx.nparams]
        saved = tensors[1 + ctx.nparams:]
        n, h, w, ct = buf.shape
        dev = buf.device
        hw, px = h * w, n * h * w
        c_in = ct - depth * k
        sd = seed_dev.data_ptr() if seed_dev is not None else None
        dbuf_in = dbuf_in.contiguous()
        dbuf = torch.empty_like(buf)                  # this node's own gradient buffer: the layers accumulate into it
        _rn.check(L.rn_dropout_strided(_rn.f32(dbuf_in), _rn.f32(dbuf), px, ct, ct, 0, ct, 0, 0.0, 0, None, _rn.stream()), "rn_dropout_strided")
        grads = [None] * ctx.nparams
        for i in range(depth - 1, -1, -1):
            g1, b1, w1, g2, b2, w2 = params[6 * i:6 * i + 6]
            a, m1, r1, d1, a2, m2, r2 = saved[7 * i:7 * i + 7]
            ci, c4 = c_in + i * k, w1.shape[3]
            dy2 = torch.empty((n, h, w, k), dtype=torch.float32, device=dev)
            _rn.check(L.rn_dropout_strided(_rn.f32(dbuf), _rn.f32(dy2), px, k, ct, ci, k, 0, rate, seeds[i][1], sd, _rn.stream()),
                      "rn_dropout_strided")
            da2, dw2 = _conv_bwd_raw(a2, w2, dy2)
            dd1 = torch.empty_like(d1)
            dg2_buf, dg2 = _grad_slot(g2)
            db2_buf, db2 = _grad_slot(b2)
            _gn_raw(False, d1, 0, c4, n, hw, g2, b2, groups, eps, act, m2, r2, dy=da2, dx=dd1, dgamma=dg2_buf, dbeta=db2_buf)
            dy1 = dd1
            if rate > 0.0:
                dy1 = torch.empty_like(dd1)
                _rn.check(L.rn_dropout(_rn.f32(dd1), _rn.f32(dy1), dd1.numel(), rate, seeds[i][0], sd, _rn.stream()), "rn_dropout")
            da, dw1 = _conv_bwd_raw(a, w1, dy1)
            dg1_buf, dg1 = _grad_slot(g1)
            db1_buf, db1 = _grad_slot(b1)
            _gn_raw(False, buf, ct, ci, n, hw, g1, b1, groups, eps, act, m1, r1, dy=da, dx=dbuf, dx_ld=ct, dx_acc=True, dgamma=dg1_buf,
                    dbeta=db1_buf)
            grads[6 * i:6 * i + 6] = [dg1, db1, dw1, dg2, db2, dw2]
        dx = torch.empty((n, h, w, c_in), dtype=torch.float32, device=dev)
        _rn.check(L.rn_dropout_strided(_rn.f32(dbuf), _rn.f32(dx), px, c_in, ct, 0, c_in, 0, 0.0, 0, None, _rn.stream()), "rn_dropout_strided")
        return (None, dx) + tuple(grads)


def dense_block(x, layers, k, groups=32, eps=1e-5, act=None, rate=0.0, seeds=None, seed_dev=None):
    """`layers`: [(gamma1, beta1, w1 [1,1,c_i,4k], gamma2, beta2, w2 [3,3,4k,k])] -> the block's [n,h,w,c_in + len(layers) k]
    output (input channels first, then every layer's k channels: the order of the reference's concat, densenet.py:119)."""
    depth = len(layers)
    seeds = tuple(seeds) if seeds is not None else tuple((0, 0) for _ in range(depth))
    cfg = (depth, int(k), int(groups), float(eps), act, float(rate), seeds, seed_dev)
    flat = [t for layer in layers for t in layer]
    return _DenseBlock.apply(cfg, x, *flat)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride):
        x = x.contiguous()
        n, h, w, c = x.shape
        oh, _ = _rn.same_pad(h, k, stride)
        ow, _ = _rn.same_pad(w, k, stride)
        y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
        # training: keep the arg-max tap of every window (1 byte per output) instead of x for the backward pass
        arg = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=x.device) if x.requires_grad else None
        _rn.check(_rn.lib().rn_maxpool_fwd(_rn.f32(x), _rn.f32(y), _rn.ptr(arg) if arg is not None else None, n, h, w, c, k,
                                           stride, _rn.stream()), "rn_maxpool_fwd")
        ctx.cfg = (k, stride, tuple(x.shape))
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        k, stride, shape = ctx.cfg
        n, h, w, c = shape
        dy = dy.contiguous()
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        _rn.check(_rn.lib().rn_maxpool_bwd_arg(_rn.ptr(arg), _rn.f32(dy), _rn.f32(dx), n, h, w, c, k, stride, _rn.stream()),
                  "rn_maxpool_bwd_arg")
        return dx, None, None


def max_pool(x, k=3, stride=2):
    """tf.layers.MaxPooling2D(k, stride, padding='same')."""
    return _MaxPool.apply(x, k, stride)


class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride):
        x = x.contiguous()
        n, h, w, c = x.shape
        oh, _ = _rn.same_pad(h, k, stride)
        ow, _ = _rn.same_pad(w, k, stride)
        y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
        _rn.check(_rn.lib().rn_avgpool_fwd(_rn.f32(x), _rn.f32(y), n, h, w, c, k, stride, _rn.stream()), "rn_avgpool_fwd")
        ctx.cfg = (k, stride, n, h, w, c)
        return y

    @staticmethod
    def backward(ctx, dy):
        k, stride, n, h, w, c = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
        _rn.check(_rn.lib().rn_avgpool_bwd(_rn.f32(dy), _rn.f32(dx), n, h, w, c, k, stride, _rn.stream()), "rn_avgpool_bwd")
        return dx, None, None


def avg_pool(x, k=2, stride=2):
    """tf.layers.AveragePooling2D(k, stride, padding='same')."""
    return _AvgPool.apply(x, k, stride)
