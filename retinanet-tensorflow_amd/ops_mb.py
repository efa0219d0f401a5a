"""The MobileNetV2 bottleneck chain as ONE autograd node over the rn_mb_* kernels (csrc/mbconv.hip): every GroupNorm of
reference mobilenet_v2.py:41-94 is applied by the kernel that consumes it, a bottleneck is 3 launches forward and 3 backward.

    taps, tail = mb_chain(x, blocks, tail_conv, tap_after)

`blocks`: one `Block` per bottleneck (expand 1x1 -> GN-act-drop -> depthwise 3x3 -> GN-act-drop -> linear 1x1 -> GN-drop
[+ input]); `tail_conv`: the 1x1 conv of the block that follows the chain (MobileNetV2's output_conv, mobilenet_v2.py:178-185)
-- it consumes the last bottleneck's GroupNorm while loading, the node returns its RAW output and the statistic rows of it.
"""
import ctypes as C
import os

import torch

import _rn
import ops


class Norm(object):
    """gamma / beta of one Normalization layer + the activation and Dropout that follow it in the reference's Sequential."""

    def __init__(self, gamma, beta, groups, eps, act, rate, seed):
        self.gamma, self.beta = gamma, beta
        self.groups_arg, self.eps, self.act, self.rate, self.seed = int(groups), float(eps), act, float(rate), int(seed)


class Block(object):
    """One bottleneck: kernels w1 [1,1,cin,wide], wd [3,3,wide,1], w3 [1,1,wide,cout], their three Norms, the stride."""

    def __init__(self, w1, n1, wd, n2, w3, n3, stride, residual):
        self.w1, self.n1, self.wd, self.n2, self.w3, self.n3 = w1, n1, wd, n2, w3, n3
        self.stride, self.residual = int(stride), bool(residual)


def _rows(query, *args):
    """(MbRows layout, bytes) of a *_rows query; bytes == 0: the shape is not supported."""
    lay = _rn.MbRows()
    nbytes = query(*args, C.byref(lay))
    return lay, int(nbytes)


COMPACT_ABOVE = int(os.environ.get('RN_MB_COMPACT_ABOVE', '256'))     # tuning aid: compact the rows of a sample when there are more
# The small-map part of the chain as phases of ONE launch per <= 14 kernels (csrc/mb_resident.hip: per-sample clusters on one XCD,
# same-XCD phase barriers).  It starts at the first kernel from which on every kernel's output map has <= RESIDENT_MAX_HW pixels
# and reads <= RESIDENT_MAX_IN_BYTES per sample (the XCD's L2 is 4 MB).  OFF by default (RN_MB_RESIDENT=1 opts in): correct
# and tested, but measured SLOWER at the headline shape -- 712 us for the 33 phases against ~420 us for the 33 launches it
# replaces (428 vs 458 images/s): a batch of two samples keeps only 2 of the 8 XCDs busy, and a phase costs what a kernel cost
# (tools/mb_resident_phases.py: ~5 us of prologue, 2 us per K-tile, 1.5 us of epilogue per tile), not the launch boundary.
RESIDENT = os.environ.get('RN_MB_RESIDENT', '0') == '1'
RESIDENT_MAX_HW = int(os.environ.get('RN_MB_RESIDENT_MAX_HW', '1024'))
RESIDENT_MAX_IN_BYTES = int(os.environ.get('RN_MB_RESIDENT_MAX_IN_BYTES', str(3328 * 1024)))


def _compact(rows, lay, n, dev):
    """(rows, layout) a consumer can merge: the producer's, or -- more rows per sample than rn_mb_rows_max() (the largest maps) --
    their sums over runs of consecutive rows (rn_mb_compact_rows, one small launch)."""
    L = _rn.lib()
    if lay.rows_per_sample <= min(L.rn_mb_rows_max(), COMPACT_ABOVE):
        return rows, lay
    src = _rn.MbRows(rows.data_ptr(), lay.rows_per_sample, lay.width, lay.bn)
    out = _rn.MbRows()
    nbytes = L.rn_mb_compact_rows_layout(n, C.byref(src), C.byref(out))
    assert nbytes
    small = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out.rows = small.data_ptr()
    _rn.check(L.rn_mb_compact_rows(C.byref(src), C.byref(out), n, _rn.stream()), "rn_mb_compact_rows")
    return small, out


class _Stage(object):
    """Buffers of one GroupNorm block inside the chain: raw tensor, statistic rows (forward), mean / rstd."""
    __slots__ = ("y", "rows", "lay", "mean", "rstd", "norm", "c", "groups", "hw")


def _mb_norm(st, training, seed_dev, with_rows):
    nm = st.norm
    m = _rn.MbNorm()
    m.y = st.y.data_ptr()
    if with_rows:
        m.stat = _rn.MbRows(st.rows.data_ptr(), st.lay.rows_per_sample, st.lay.width, st.lay.bn)
    m.mean, m.rstd = st.mean.data_ptr(), st.rstd.data_ptr()
    m.gamma, m.beta = _rn.f32(nm.gamma), _rn.f32(nm.beta)
    m.c, m.groups, m.act, m.eps = st.c, st.groups, _rn.ACT[nm.act], nm.eps
    rate = nm.rate if training else 0.0
    m.drop_rate, m.drop_seed = rate, nm.seed
    m.drop_seed_dev = seed_dev.data_ptr() if (seed_dev is not None and rate > 0.0) else None
    return m


def _count_resident(nphases):
    """(test hook: called with the number of phases whenever a forward pass takes the resident section)"""


def _resident_start(n, h, w, c, blocks_cfg, shapes, tail_cout):
    """Index of the first kernel (3 i + j for kernel j of bottleneck i, 3 nb for the tail conv) of the resident section, or
    None.  `shapes`: per bottleneck (wide, cout).  Every kernel from the index on must qualify (maps only shrink along the chain)."""
    if not RESIDENT:
        return None
    L = _rn.lib()
    ok = []
    for (stride, _res, n1, n2, n3), (wide, cout) in zip(blocks_cfg, shapes):
        oh, _ = _rn.same_pad(h, 3, stride)
        ow, _ = _rn.same_pad(w, 3, stride)
        g1, g3 = ops.gn_groups(wide, n1.groups_arg), ops.gn_groups(cout, n3.groups_arg)
        acts = (n1.act in ('elu',), n2.act in ('elu',), n3.act is None)
        ok.append(h * w <= RESIDENT_MAX_HW and h * w * c * 4 <= RESIDENT_MAX_IN_BYTES and acts[2] and
                  bool(L.rn_mb_resident_rows(_rn.MB_PHASE_POINTWISE, n, h, w, c, wide, 1, g1, None)))
        ok.append(oh * ow <= RESIDENT_MAX_HW and h * w * wide * 4 <= RESIDENT_MAX_IN_BYTES and acts[0] and
                  bool(L.rn_mb_resident_rows(_rn.MB_PHASE_DEPTHWISE, n, h, w, wide, wide, stride, g1, None)))
        ok.append(oh * ow <= RESIDENT_MAX_HW and oh * ow * wide * 4 <= RESIDENT_MAX_IN_BYTES and acts[1] and
                  bool(L.rn_mb_resident_rows(_rn.MB_PHASE_POINTWISE, n, oh, ow, wide, cout, 1, g3, None)))
        h, w, c = oh, ow, cout
    ok.append(h * w <= RESIDENT_MAX_HW and h * w * c * 4 <= RESIDENT_MAX_IN_BYTES and h * w % 32 == 0 and c % 4 == 0 and tail_cout % 4 == 0)
    ok[0] = False                   # the chain's first kernel reads a plain tensor: launch-ordered
    start = len(ok)
    while start > 0 and ok[start - 1]:
        start -= 1
    return start if start < len(ok) else None


def chain_supported(shape, blocks):
    """Can the rn_mb_* kernels run `blocks` on an fp32 device tensor of `shape` [n,h,w,c]?  (Every row layout must exist.)"""
    L = _rn.lib()
    n, h, w, c = (int(v) for v in shape)
    for b in blocks:
        wide, cout = b.w1.shape[3], b.w3.shape[3]
        if b.w1.shape[2] != c or wide % 4 or cout % 4 or c % 4:
            return False
        g1, g3 = ops.gn_groups(wide, b.n1.groups_arg), ops.gn_groups(cout, b.n3.groups_arg)
        oh, _ = _rn.same_pad(h, 3, b.stride)
        ow, _ = _rn.same_pad(w, 3, b.stride)
        if not L.rn_mb_pointwise_rows(n, h * w, c, wide, g1, None) or not L.rn_mb_depthwise_rows(n, h, w, wide, b.stride, g1, None):
            return False
        if not L.rn_mb_pointwise_rows(n, oh * ow, wide, cout, g3, None):
            return False
        if not L.rn_mb_pointwise_bwd_rows(n, oh * ow, wide, cout, g1, None) or not L.rn_mb_depthwise_bwd_rows(n, h, w, wide, b.stride, g1, None):
            return False
        if not L.rn_mb_pointwise_bwd_rows(n, h * w, c, wide, ops.gn_groups(c, 32), None):
            return False
        h, w, c = oh, ow, cout
    return True


class _MbChain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, x, *params):
        blocks_cfg, tail_cfg, tap_after, training, seed_dev = cfg
        L = _rn.lib()
        dev = x.device
        x = x.contiguous()
        n, h, w, c = x.shape
        nb = len(blocks_cfg)
        st_ = _rn.stream()

        shapes = [(params[9 * i].shape[3], params[9 * i + 6].shape[3]) for i in range(nb)]
        res_from = _resident_start(n, h, w, c, blocks_cfg, shapes, params[9 * nb].shape[3]) if x.is_cuda else None
        phases, keep = [], []       # the resident section's phases (rn_mb_phase) and what they point to

        def resident(k):
            return res_from is not None and k >= res_from

        def phase(kind, nm, res_t, mat_t, w_t, y_t, hh, ww, cin, cout, stride, s_out, groups):
            ph = _rn.MbPhase()
            ph.kind, ph.in_ = kind, C.pointer(nm)
            ph.residual = _rn.f32(res_t) if res_t is not None else None
            ph.materialise = _rn.f32(mat_t) if mat_t is not None else None
            ph.w, ph.y = _rn.f32(w_t), _rn.f32(y_t)
            ph.h, ph.wd, ph.cin, ph.cout, ph.stride = hh, ww, cin, cout, stride
            if s_out is not None:
                ph.stat_out = _rn.MbRows(s_out.rows.data_ptr(), s_out.lay.rows_per_sample, s_out.lay.width, s_out.lay.bn)
                ph.stat_groups = groups
            keep.append(nm)
            phases.append(ph)

        def new_stage(y, query_args, query, groups, norm, res_kind=None, res_args=None):
            s = _Stage()
            s.y, s.c, s.groups, s.norm, s.hw = y, y.shape[3], groups, norm, y.shape[1] * y.shape[2]
            if res_kind is not None:        # a resident phase writes its rows in the cluster's tiling
                s.lay, nbytes = _rows(L.rn_mb_resident_rows, res_kind, *res_args)
            else:
                s.lay, nbytes = _rows(query, *query_args)
            assert nbytes, "mb_chain: unsupported shape (chain_supported() says so)"
            s.rows = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            s.mean = torch.empty((y.shape[0], groups), dtype=torch.float32, device=dev)
            s.rstd = torch.empty((y.shape[0], groups), dtype=torch.float32, device=dev)
            return s

        def rows_arg(s):
            return C.byref(_rn.MbRows(s.rows.data_ptr(), s.lay.rows_per_sample, s.lay.width, s.lay.bn))

        saved = []                  # per block: (x_in, stage1, stage2, stage3)
        taps = []
        pend, pend_res = None, None   # the GroupNorm block whose output is the next kernel's A operand, and its residual
        x_in = x
        pi = 0
        for i, bc in enumerate(blocks_cfg):
            stride, residual, n1, n2, n3 = bc
            w1, g1_, b1_, wd, g2_, b2_, w3, g3_, b3_ = params[pi:pi + 9]
            pi += 9
            n1.gamma, n1.beta, n2.gamma, n2.beta, n3.gamma, n3.beta = g1_, b1_, g2_, b2_, g3_, b3_
            wide, cout = w1.shape[3], w3.shape[3]
            gr1, gr3 = ops.gn_groups(wide, n1.groups_arg), ops.gn_groups(cout, n3.groups_arg)
            oh, _ = _rn.same_pad(h, 3, stride)
            ow, _ = _rn.same_pad(w, 3, stride)
            # expand 1x1: A = the chain input, or the previous bottleneck's output formed while loading (and written out once)
            y1 = torch.empty((n, h, w, wide), dtype=torch.float32, device=dev)
            r0, r1, r2 = resident(3 * i), resident(3 * i + 1), resident(3 * i + 2)
            s1 = new_stage(y1, (n, h * w, c, wide, gr1), L.rn_mb_pointwise_rows, gr1, n1,
                           _rn.MB_PHASE_POINTWISE if r0 else None, (n, h, w, c, wide, 1, gr1))
            if r0:
                x_in = torch.empty((n, h, w, c), dtype=torch.float32, device=dev)
                phase(_rn.MB_PHASE_POINTWISE, _mb_norm(pend, training, seed_dev, True), pend_res, x_in, w1, y1, h, w, c, wide, 1, s1, gr1)
                if (i - 1) in tap_after:
                    taps.append(x_in)
            elif pend is None:
                _rn.check(L.rn_mb_pointwise_fwd(_rn.f32(x_in), None, None, None, _rn.f32(w1), _rn.f32(y1), n, h * w, c, wide, rows_arg(s1), gr1, st_),
                          "rn_mb_pointwise_fwd")
            else:
                x_in = torch.empty((n, h, w, c), dtype=torch.float32, device=dev)
                nm = _mb_norm(pend, training, seed_dev, True)
                _rn.check(L.rn_mb_pointwise_fwd(None, C.byref(nm), _rn.f32(pend_res) if pend_res is not None else None, _rn.f32(x_in),
                                                _rn.f32(w1), _rn.f32(y1), n, h * w, c, wide, rows_arg(s1), gr1, st_), "rn_mb_pointwise_fwd")
                if (i - 1) in tap_after:
                    taps.append(x_in)
            if not r0:
                s1.rows, s1.lay = _compact(s1.rows, s1.lay, n, dev)
            # depthwise 3x3 on drop(act(GN1(y1)))
            y2 = torch.empty((n, oh, ow, wide), dtype=torch.float32, device=dev)
            s2 = new_stage(y2, (n, h, w, wide, stride, gr1), L.rn_mb_depthwise_rows, gr1, n2,
                           _rn.MB_PHASE_DEPTHWISE if r1 else None, (n, h, w, wide, wide, stride, gr1))
            nm = _mb_norm(s1, training, seed_dev, True)
            if r1:
                phase(_rn.MB_PHASE_DEPTHWISE, nm, None, None, wd, y2, h, w, wide, wide, stride, s2, gr1)
            else:
                _rn.check(L.rn_mb_depthwise_fwd(C.byref(nm), _rn.f32(wd), _rn.f32(y2), n, h, w, stride, rows_arg(s2), gr1, st_), "rn_mb_depthwise_fwd")
                s2.rows, s2.lay = _compact(s2.rows, s2.lay, n, dev)
            # linear 1x1 on drop(act(GN2(y2)))
            y3 = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=dev)
            s3 = new_stage(y3, (n, oh * ow, wide, cout, gr3), L.rn_mb_pointwise_rows, gr3, n3,
                           _rn.MB_PHASE_POINTWISE if r2 else None, (n, oh, ow, wide, cout, 1, gr3))
            nm = _mb_norm(s2, training, seed_dev, True)
            if r2:
                phase(_rn.MB_PHASE_POINTWISE, nm, None, None, w3, y3, oh, ow, wide, cout, 1, s3, gr3)
            else:
                _rn.check(L.rn_mb_pointwise_fwd(None, C.byref(nm), None, None, _rn.f32(w3), _rn.f32(y3), n, oh * ow, wide, cout, rows_arg(s3), gr3, st_),
                          "rn_mb_pointwise_fwd")
                s3.rows, s3.lay = _compact(s3.rows, s3.lay, n, dev)
            saved.append((x_in, s1, s2, s3))
            pend, pend_res = s3, (x_in if residual else None)
            h, w, c = oh, ow, cout
        # the conv after the chain consumes the last GroupNorm (and writes the last bottleneck's output out: its weight gradient reads it)
        tail_w = params[pi]
        tail_groups = tail_cfg
        ct = tail_w.shape[3]
        x_last = torch.empty((n, h, w, c), dtype=torch.float32, device=dev)
        y_t = torch.empty((n, h, w, ct), dtype=torch.float32, device=dev)
        nm = _mb_norm(pend, training, seed_dev, True)
        # the tail's rows go out in the layer-by-layer GroupNorm's layout: a stand-alone apply follows (ops.group_norm_act)
        if resident(3 * nb):
            phase(_rn.MB_PHASE_POINTWISE, nm, pend_res, x_last, tail_w, y_t, h, w, c, ct, 1, None, 0)
        else:
            assert not phases, "the resident section runs to the end of the chain"
            _rn.check(L.rn_mb_pointwise_fwd(None, C.byref(nm), _rn.f32(pend_res) if pend_res is not None else None, _rn.f32(x_last), _rn.f32(tail_w),
                                            _rn.f32(y_t), n, h * w, c, ct, None, 0, st_), "rn_mb_pointwise_fwd")
        if phases:
            _count_resident(len(phases))
            arr = (_rn.MbPhase * len(phases))(*phases)
            _rn.check(L.rn_mb_resident_fwd(arr, len(phases), n, _rn.resident_sync(dev).data_ptr(), st_), "rn_mb_resident_fwd")
        if (nb - 1) in tap_after:
            taps.append(x_last)
        ctx.cfg = cfg
        ctx.saved = saved
        ctx.x_last = x_last
        ctx.params = params
        ctx.shape_in = tuple(x.shape)
        ctx.set_materialize_grads(False)
        return tuple(taps) + (y_t,)

    @staticmethod
    def backward(ctx, *grads):
        blocks_cfg, tail_cfg, tap_after, training, seed_dev = ctx.cfg
        L = _rn.lib()
        params = ctx.params
        saved = ctx.saved
        nb = len(blocks_cfg)
        dev = ctx.x_last.device
        st_ = _rn.stream()
        taps_sorted = sorted(tap_after)
        tap_grad = {t: grads[k] for k, t in enumerate(taps_sorted)}
        dy_t = grads[-1]
        pgrads = [None] * len(params)
        n = ctx.shape_in[0]

        def grows_for(query, args, c):
            lay, nbytes = _rows(query, *args)
            assert nbytes
            rows = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            planes = torch.empty((2, n * lay.rows_per_sample, c), dtype=torch.float32, device=dev)
            if ops._deferring:
                ops._deferred_keep.append(planes)
            return lay, rows, planes

        def norm_param_grads(planes, nm, slot_g, slot_b):
            """dbeta | dgamma = column sums of the planes (joins the step's deferred reduction)."""
            c, nrows = planes.shape[2], planes.shape[1]
            db_buf, db = ops._grad_slot(nm.beta)
            dg_buf, dg = ops._grad_slot(nm.gamma)
            _rn.check(L.rn_reduce_rows(_rn.f32(planes[0]), _rn.f32(db_buf), c, nrows, 0, st_, ops._defer_arg()), "rn_reduce_rows")
            _rn.check(L.rn_reduce_rows(_rn.f32(planes[1]), _rn.f32(dg_buf), c, nrows, 0, st_, ops._defer_arg()), "rn_reduce_rows")
            pgrads[slot_g], pgrads[slot_b] = dg, db

        def pw_bwd(x_plain, in_stage, dy, w, gout, hw, cin, cout, const_w=False):
            if const_w:             # a fixed kernel (the identity of a stage cut): its weight gradient goes nowhere
                dw_buf, dw = torch.empty_like(w), None
                if ops._deferring:
                    ops._deferred_keep.append(dw_buf)
            else:
                dw_buf, dw = ops._grad_slot(w)
            need = L.rn_mb_pointwise_bwd_workspace(n, hw, cin, cout)
            ws = ops._grad_workspace(need, dev)
            nm_in = _mb_norm(in_stage, training, seed_dev, False) if in_stage is not None else None
            _rn.check(L.rn_mb_pointwise_bwd(_rn.f32(x_plain) if x_plain is not None else None, C.byref(nm_in) if nm_in is not None else None,
                                            C.byref(dy), _rn.f32(w), _rn.f32(dw_buf), C.byref(gout), n, hw, cin, cout, ws.data_ptr(), ws.numel(),
                                            st_, ops._defer_arg()), "rn_mb_pointwise_bwd")
            return dw

        # ---- the tail conv: plain dy; its data gradient enters the last bottleneck's GroupNorm 3
        x_last = ctx.x_last
        _, hl, wl, cl = x_last.shape
        tail_w = params[9 * nb]
        ct = tail_w.shape[3]
        if dy_t is None:
            dy_t = torch.zeros((n, hl, wl, ct), dtype=torch.float32, device=dev)
        dy_t = dy_t.contiguous()
        s3_last = saved[-1][3]
        lay3, rows3, planes3 = grows_for(L.rn_mb_pointwise_bwd_rows, (n, hl * wl, cl, ct, s3_last.groups), cl)
        D = torch.empty((n, hl, wl, cl), dtype=torch.float32, device=dev)
        nm3 = _mb_norm(s3_last, training, seed_dev, False)
        tg = tap_grad.get(nb - 1)
        tg = tg.contiguous() if tg is not None else None      # (kept alive in `keep` until the chain's last launch is queued)
        gout = _rn.MbGout(D.data_ptr(), None, _rn.f32(tg) if tg is not None else None, C.pointer(nm3), 1,
                          _rn.MbRows(rows3.data_ptr(), lay3.rows_per_sample, lay3.width, lay3.bn), planes3.data_ptr())
        dyd = _rn.MbDy(dy_t.data_ptr(), None, None, 0, _rn.MbRows())
        pgrads[9 * nb] = pw_bwd(x_last, None, dyd, tail_w, gout, hl * wl, cl, ct, const_w=bool(tail_cfg))
        rows3, lay3 = _compact(rows3, lay3, n, dev)
        keep = [tg]

        dx0 = None
        flushed = False
        for i in range(nb - 1, -1, -1):
            stride, residual, n1, n2, n3 = blocks_cfg[i]
            x_in, s1, s2, s3 = saved[i]
            if not flushed and saved[i][0].shape[1] * saved[i][0].shape[2] >= 16384:
                # the large maps begin: everything recorded so far is reduced beside them (ops.flush_deferred_midway)
                ops.flush_deferred_midway(dev)
                flushed = True
            w1, g1_, b1_, wd, g2_, b2_, w3, g3_, b3_ = params[9 * i:9 * i + 9]
            _, h, w, c = x_in.shape
            wide, cout = w1.shape[3], w3.shape[3]
            _, oh, ow, _ = s2.y.shape
            # (D, rows3, planes3): gradient of this bottleneck's output and its GroupNorm-3 sums, from the kernel behind it
            norm_param_grads(planes3, n3, 9 * i + 7, 9 * i + 8)
            # linear conv backward: dy3 formed from D (mask on load); data gradient -> g2 of GroupNorm 2
            nm3 = _mb_norm(s3, training, seed_dev, False)
            nm2 = _mb_norm(s2, training, seed_dev, False)
            dy3 = _rn.MbDy(None, C.pointer(nm3), D.data_ptr(), 1, _rn.MbRows(rows3.data_ptr(), lay3.rows_per_sample, lay3.width, lay3.bn))
            lay2, rows2, planes2 = grows_for(L.rn_mb_pointwise_bwd_rows, (n, oh * ow, wide, cout, s2.groups), wide)
            g2 = torch.empty_like(s2.y)
            gout = _rn.MbGout(g2.data_ptr(), None, None, C.pointer(nm2), 0,
                              _rn.MbRows(rows2.data_ptr(), lay2.rows_per_sample, lay2.width, lay2.bn), planes2.data_ptr())
            pgrads[9 * i + 6] = pw_bwd(None, s2, dy3, w3, gout, oh * ow, wide, cout)
            rows2, lay2 = _compact(rows2, lay2, n, dev)
            norm_param_grads(planes2, n2, 9 * i + 4, 9 * i + 5)
            # depthwise backward: dy2 from g2; data gradient -> g1 of GroupNorm 1
            nm1 = _mb_norm(s1, training, seed_dev, False)
            dy2 = _rn.MbDy(None, C.pointer(nm2), g2.data_ptr(), 0, _rn.MbRows(rows2.data_ptr(), lay2.rows_per_sample, lay2.width, lay2.bn))
            lay1, rows1, planes1 = grows_for(L.rn_mb_depthwise_bwd_rows, (n, h, w, wide, stride, s1.groups), wide)
            g1 = torch.empty_like(s1.y)
            gout = _rn.MbGout(g1.data_ptr(), None, None, C.pointer(nm1), 0,
                              _rn.MbRows(rows1.data_ptr(), lay1.rows_per_sample, lay1.width, lay1.bn), planes1.data_ptr())
            dwd_buf, dwd = ops._grad_slot(wd)
            need = L.rn_mb_depthwise_bwd_workspace(n, h, w, wide, stride)
            ws = ops._grad_workspace(need, dev)
            _rn.check(L.rn_mb_depthwise_bwd(C.byref(nm1), C.byref(dy2), _rn.f32(wd), _rn.f32(dwd_buf), C.byref(gout), n, h, w, stride,
                                            ws.data_ptr(), ws.numel(), st_, ops._defer_arg()), "rn_mb_depthwise_bwd")
            pgrads[9 * i + 3] = dwd
            rows1, lay1 = _compact(rows1, lay1, n, dev)
            norm_param_grads(planes1, n1, 9 * i + 1, 9 * i + 2)
            # expand conv backward: dy1 from g1; data gradient (+ the residual path's D, + a tap's gradient) enters the previous
            # bottleneck's GroupNorm 3 -- or leaves the chain
            dy1 = _rn.MbDy(None, C.pointer(nm1), g1.data_ptr(), 0, _rn.MbRows(rows1.data_ptr(), lay1.rows_per_sample, lay1.width, lay1.bn))
            Dn = torch.empty((n, h, w, c), dtype=torch.float32, device=dev)
            add1 = D.data_ptr() if residual else None
            if i > 0:
                s3p = saved[i - 1][3]
                lay3n, rows3n, planes3n = grows_for(L.rn_mb_pointwise_bwd_rows, (n, h * w, c, wide, s3p.groups), c)
                nm3p = _mb_norm(s3p, training, seed_dev, False)
                tg = tap_grad.get(i - 1)
                tg = tg.contiguous() if tg is not None else None
                keep.append(tg)
                gout = _rn.MbGout(Dn.data_ptr(), add1, _rn.f32(tg) if tg is not None else None, C.pointer(nm3p), 1,
                                  _rn.MbRows(rows3n.data_ptr(), lay3n.rows_per_sample, lay3n.width, lay3n.bn), planes3n.data_ptr())
            else:
                gout = _rn.MbGout(Dn.data_ptr(), add1, None, None, 1, _rn.MbRows(), None)
            pgrads[9 * i] = pw_bwd(x_in, None, dy1, w1, gout, h * w, c, wide)
            if i > 0:
                rows3n, lay3n = _compact(rows3n, lay3n, n, dev)
                lay3, rows3, planes3 = lay3n, rows3n, planes3n
            else:
                dx0 = Dn
            D_prev, D = D, Dn
            del D_prev
        return (None, dx0) + tuple(pgrads)


def mb_chain(x, blocks, tail_w, tap_after=(), training=True, seed_dev=None, tail_const=False):
    """x [n,h,w,c] -> (list of tap tensors (outputs of the bottlenecks whose index is in `tap_after`, ascending), raw output of
    the tail 1x1 conv applied to the last bottleneck's output).  See the module docstring.  tail_const: `tail_w` is a fixed
    kernel, not a parameter (the identity with which a stage cut ends the first half of a chain, mobilenet_v2.py): no weight
    gradient is returned for it."""
    cfg_blocks = tuple((b.stride, b.residual, b.n1, b.n2, b.n3) for b in blocks)
    flat = []
    for b in blocks:
        flat += [b.w1, b.n1.gamma, b.n1.beta, b.wd, b.n2.gamma, b.n2.beta, b.w3, b.n3.gamma, b.n3.beta]
    flat.append(tail_w)
    cfg = (cfg_blocks, bool(tail_const), frozenset(tap_after), bool(training), seed_dev)
    out = _MbChain.apply(cfg, x, *flat)
    return list(out[:-1]), out[-1]
