"""Layer containers with the reference's calling convention (reference model.py:5-25).

``Model`` is the base class of every block; ``Sequential(layers)(input, training)`` runs the
layers in order and hands ``training=`` only to callees whose signature has it
(model.py:16-25).  Device work happens in the layers (hand-written HIP kernels via ops.py);
this file is host-side plumbing.  On top of the reference behaviour, ``Sequential`` fuses
the ubiquitous ``[Normalization, activation, Dropout]`` run into ONE GroupNorm kernel and
accepts a list of tensors (pyramid levels sharing the layers) that run as one launch per
layer.
"""
import inspect

import torch

import layers as L


class Model(torch.nn.Module):
    def __init__(self, name='model'):
        super().__init__()
        self.model_name = name

    def call(self, *args, **kwargs):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        return self.call(*args, **kwargs)


def _accepts_training(fn):
    try:
        return 'training' in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


class Sequential(Model):
    def __init__(self, layers, name='sequential'):
        super().__init__(name=name)
        self.layers = list(layers)
        # register sub-modules so parameters are found; plain callables (activations) stay as is
        self._mods = torch.nn.ModuleList([l for l in self.layers if isinstance(l, torch.nn.Module)])

    def call(self, input, training, residual=None):
        """`residual` (optional) is added after the LAST layer; it is fused into the GroupNorm
        kernel when the run ends with a fusable [Normalization, act, Dropout] group."""
        i, n = 0, len(self.layers)
        while i < n:
            layer = self.layers[i]
            if isinstance(layer, L.GroupNormalization):
                act, drop, j = None, None, i + 1
                if j < n and L.activation_name(self.layers[j]) is not None:
                    act, j = L.activation_name(self.layers[j]), j + 1
                if j < n and isinstance(self.layers[j], L.Dropout):
                    drop, j = self.layers[j], j + 1
                res = residual if j == n else None
                input = layer.fused(input, training, act=act, dropout=drop, residual=res)
                if res is not None:
                    residual = None
                i = j
                continue
            if (isinstance(layer, (L.Conv2D, L.DepthwiseConv2D)) and i + 1 < n and isinstance(self.layers[i + 1], L.GroupNormalization)
                    and torch.is_tensor(input) and input.dtype == torch.float32 and not (L.INFERENCE_F16 and not training)):
                input = layer(input, norm=self.layers[i + 1])    # the conv also emits the GroupNorm's statistics where it can
                i += 1
                continue
            if (L.INFERENCE_F16 and not training and isinstance(layer, L.Conv2D) and i + 1 < n
                    and isinstance(self.layers[i + 1], L.GroupNormalization) and torch.is_tensor(input) and input.is_cuda
                    and input.dtype == torch.float16 and layer.weight is not None and layer.bias is None):
                # fp16 inference: the GroupNorm's statistics from the conv's epilogue (ops_f16.conv2d_norm) -- no statistics pass
                # over the conv output; the GroupNorm [+ activation] is then ONE apply pass.  Dropout is the identity here.
                import ops_f16
                j, act = i + 2, None
                if j < n and L.activation_name(self.layers[j]) is not None:
                    act, j = L.activation_name(self.layers[j]), j + 1
                if j < n and isinstance(self.layers[j], L.Dropout):
                    j += 1
                p = ops_f16.conv2d_norm(input, layer.weight, self.layers[i + 1], act, layer.strides, layer.groups) if ops_f16.FOLD else None
                if p is not None:
                    res = residual if j == n else None
                    input = p.materialise(residual=res)
                    if res is not None:
                        residual = None
                    i = j
                    continue
            target = layer.call if isinstance(layer, Model) else layer
            if isinstance(layer, torch.nn.Module) and not isinstance(layer, Model):
                target = layer.forward
            if _accepts_training(target):
                input = layer(input, training=training)
            else:
                input = layer(input)
            i += 1
        if residual is not None:
            input = L.add(input, residual)
        return input
