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
import os

import numpy as np
import torch

import _rn
import layers as L
import losses
import ops
import utils
from levels import build_levels

OPT_BLOCK = _rn.OPT_BLOCK


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
        self.step_count = 0

    def step(self, grad_scale=1.0):
        a = self.arena
        L_ = _rn.lib()
        ws = _rn.workspace(L_.rn_optimizer_workspace(a.count), a.weights.device)
        _rn.check(L_.rn_grad_norm_l2reg(_rn.f32(a.weights), _rn.f32(a.grads), _rn.f32(a.wd_per_block), a.count,
                                        grad_scale, _rn.f32(self.norm_reg), ws.data_ptr(), ws.numel(), _rn.stream()),
                  'rn_grad_norm_l2reg')
        self.step_count += 1
        _rn.check(L_.rn_optimizer_step(_rn.OPT[self.kind], _rn.f32(a.weights), _rn.f32(a.grads), _rn.f32(self.state1),
                                       _rn.f32(self.state2) if self.state2 is not None else None,
                                       _rn.f32(a.wd_per_block), a.count, self.lr, grad_scale, self.clip,
                                       _rn.f32(self.norm_reg), self.step_count, _rn.stream()), 'rn_optimizer_step')

    @property
    def regularization_loss(self):
        """sum_w scale * ||w||^2 / 2 at the weights used for the last gradient (train.py:221)."""
        return self.norm_reg[1]


class GradientAllReduce(object):
    """MirroredStrategy's cross-replica gradient sum (train.py:261-267) as RCCL all-reduces of
    contiguous arena slices.  xGMI is point-to-point (7 links x ~153 GB/s), ring collectives are
    per-link bound, so the arena goes out as FEW LARGE buckets (default 64 MB: the whole 40 MB arena of the
    headline config is ONE message) rather than one
    message per tensor; the 1/world_size average is folded into the optimizer kernel."""

    def __init__(self, arena, process_group=None, bucket_bytes=64 << 20):
        import torch.distributed as dist
        self.dist, self.group, self.arena = dist, process_group, arena
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        n = arena.count
        per = max(OPT_BLOCK, (bucket_bytes // 4) // OPT_BLOCK * OPT_BLOCK)
        self.buckets = [(s, min(s + per, n)) for s in range(0, n, per)]

    def __call__(self):
        if self.world == 1:
            return 1.0
        # reverse order: the heads' gradients (end of the arena, 64 % of the bytes) are complete first
        works = [self.dist.all_reduce(self.arena.grads[s:e], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for s, e in reversed(self.buckets)]
        for w in works:
            w.wait()
        return 1.0 / self.world


class Trainer(object):
    """One data-parallel replica.  `step(features)` = forward + loss + backward + gradient
    all-reduce + optimizer.  With use_graph=True the replica-local part is one hipGraph."""

    def __init__(self, net, levels=None, optimizer='momentum', learning_rate=1e-2, grad_clip_norm=None,
                 loss_mode='bce_dice', device='cuda', use_graph=False, process_group=None,
                 direct_param_grads=True, wgrad_side_stream=False, defer_reductions=True):
        self.net, self.levels = net, levels or build_levels()
        self.device = torch.device(device)
        self.loss_mode = loss_mode
        self.arena = ParamArena(net, self.device)
        self.opt = Optimizer(self.arena, optimizer, learning_rate, grad_clip_norm)
        self.allreduce = GradientAllReduce(self.arena, process_group)
        self.use_graph = use_graph
        self.defer_reductions = (bool(defer_reductions) and bool(direct_param_grads) and
                                 os.environ.get("RN_DEFER_REDUCTIONS", "1") == "1")
        # kernels write parameter gradients straight into the arena (every parameter of this network
        # is used by exactly one op call per step); see ops.DIRECT_PARAM_GRADS
        ops.DIRECT_PARAM_GRADS = bool(direct_param_grads)
        # ... which would let the weight-gradient kernels run on a side stream under the dgrad / GroupNorm chain;
        # measured on MI355X: 162 vs 170 img/s (fork/join edges + CU contention cost more than the overlap buys), so off
        ops.WGRAD_SIDE_STREAM = bool(direct_param_grads) and bool(wgrad_side_stream)
        self._graph = None
        self._static = None
        self.drop_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._one = torch.ones((), dtype=torch.float32, device=self.device)
        L.Dropout.seed_device_counter = self.drop_counter
        self.last = {}

    # -- replica-local work (capturable)
    def forward_backward(self, features):
        logits = {'detection': self.net(features['image'], training=True)}
        inp, logits = utils.process_labels_and_logits(labels=features, logits=logits, levels=self.levels)
        class_loss, regr_loss = losses.loss(labels=inp['detection_trainable'], logits=logits['detection_trainable'],
                                            mode=self.loss_mode)
        # kernels that write a parameter's gradient directly overwrite it; gradients that reach a parameter
        # through autograd (e.g. the concatenated head kernels) are accumulated -> the arena starts at zero
        self.arena.zero_grad()
        defer = self.defer_reductions and self.device.type == 'cuda' and ops.DIRECT_PARAM_GRADS
        if defer:                                  # ~130 gradient row reductions -> one launch after backward
            import retinanet
            extra = [retinanet.side_stream(self.device)] if retinanet.HEADS_TWO_STREAMS else []
            if ops.WGRAD_SIDE_STREAM:
                extra.append(_rn.side_stream(self.device, 0))
            if os.environ.get("RN_DEFER_SIDE", "1") != "1":
                extra = []
            ops.begin_deferred_reductions(extra)
        try:
            # d(class_loss + regr_loss): both roots seeded with the same pre-allocated 1 (no add / fill kernels in the step)
            torch.autograd.backward([class_loss, regr_loss], [self._one, self._one])
        finally:
            if defer:
                ops.end_deferred_reductions()
        # weight-gradient kernels run on a side stream (ops.WGRAD_SIDE_STREAM) and write straight into the
        # arena: join every side stream before anything reads the arena
        if self.device.type == 'cuda':
            _rn.join_side_streams(self.device)
        self.drop_counter += 0x9E3779B9            # fresh dropout masks next step (device-side counter)
        return class_loss.detach(), regr_loss.detach()

    def _run_local(self, features):
        if not self.use_graph:
            return self.forward_backward(features)
        if self._graph is None:
            self._static = features
            # warm-up on a side stream (allocator + workspace sizing), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    self.forward_backward(self._static)
            torch.cuda.current_stream().wait_stream(s)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                self._graph_out = self.forward_backward(self._static)
        else:
            _copy_tree(self._static, features)
        self._graph.replay()
        return self._graph_out

    def step(self, features):
        class_loss, regr_loss = self._run_local(features)
        grad_scale = self.allreduce()
        self.opt.step(grad_scale)
        self.last = {'class_loss': class_loss, 'regr_loss': regr_loss,
                     'regularization_loss': self.opt.regularization_loss}
        return self.last


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
    return parser


def main(argv=None):
    import checkpoint
    import dataset
    import retinanet
    args = build_parser().parse_args(argv)
    assert args.dataset[0] == 'shapes', 'only the synthetic shapes loader is built in (file readers are out of scope)'
    dev = torch.device('cuda', int(__import__('os').environ.get('LOCAL_RANK', '0')))
    from data_loaders.shapes import Shapes
    loader = Shapes(None, image_size=(args.scale + args.scale // 4, args.scale))   # rescale_image brings it to --scale
    levels = build_levels()
    net = retinanet.RetinaNet(backbone=args.backbone, levels=levels, num_classes=loader.num_classes, activation=L.elu,
                              dropout_rate=args.dropout).to(dev)
    trainer = Trainer(net, levels, optimizer=args.optimizer, learning_rate=args.learning_rate,
                      grad_clip_norm=args.grad_clip_norm, loss_mode=args.loss, device=dev)
    step = 0
    path = None if args.experiment is None else __import__('os').path.join(args.experiment, 'model.safetensors')
    if path is not None and __import__('os').path.exists(path):
        step = checkpoint.load(path, net, trainer)
        print('restored step', step)
    it = dataset.build_dataset(loader, levels, scale=args.scale, device=dev)       # train.py:192-203 train_input_fn
    for epoch in range(args.epochs):
        for _ in range(args.steps_per_epoch):
            out = trainer.step(next(it))                                           # batch = [image, hflip]
            step += 1
            if step % 20 == 0:
                print('epoch %d step %d class_loss %.4f regr_loss %.4f reg %.4f' % (
                    epoch, step, out['class_loss'].item(), out['regr_loss'].item(), out['regularization_loss'].item()),
                    flush=True)
        if path is not None:
            checkpoint.save(path, net, trainer, step=step)
    if args.eval_images:
        res = evaluate(net, Shapes(None, image_size=(args.scale + args.scale // 4, args.scale), seed=12345), levels,
                       args.eval_images, scale=args.scale, device=dev)
        print('eval: mAP %.4f AP50 %.4f AP75 %.4f class_iou %.4f regr_iou %.4f over %d images' % (
            res['mAP'], res['AP50'], res['AP75'], res['class_iou'], res['regr_iou'], res['images']), flush=True)
    return step


if __name__ == '__main__':
    main()
