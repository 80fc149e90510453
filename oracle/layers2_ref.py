"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch autograd, any float dtype) of the gen-2 layer forms that the gen-1
models never use: instance norm, batch renorm and the two-conv residual block.  Only tests/ may import this file.

Parity status: UNPINNED (TensorFlow is not installed here and the reference holds no fixtures for these layers); the
restatement follows the reference's source text:
  * instance_norm      hem/ops/images.py:73-89   moments over H, W per image and channel (biased variance), eps 1e-3,
                                                 then scale (ones) * normalised + shift (zeros)
  * conv2d / deconv2d  hem/ops/layers.py:121-130, 197-206   conv + bias, [instance norm], [batch norm], activation
  * residual           hem/ops/layers.py:215-320
  * batch renorm       hem/ops/layers.py:62 (`renorm=use_batch_renorm`): tf.contrib.layers.batch_norm in training mode with
                       renorm moving averages that are still at their zero initial values (the update ops never run,
                       SURVEY.md App. C-3), so r = 1 and d = 0 and the layer is `batch_norm` below.
Tensors are NHWC; filters HWIO (deconv: [k, k, Cout, Cin]).
"""
import torch

from . import torch_ref as TR

IN_EPS = 1e-3          # hem/ops/images.py:83


def instance_norm(x, scale, shift):
    """hem/ops/images.py:73-89."""
    mu = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(1, 2), keepdim=True)          # tf.nn.moments: biased
    return scale * ((x - mu) / (var + IN_EPS) ** 0.5) + shift


def batch_renorm_initial(x, beta):
    """Batch renorm with untouched (zero) renorm statistics == training-mode batch norm (see the module docstring)."""
    return TR.batch_norm(x, beta)


def conv2d(x, P, scope, name, stride, act, norm=None, bn_beta=None, padding='SAME'):
    """hem/ops/layers.py:70-135.  norm in (None, 'instance', 'batch', 'renorm')."""
    conv = TR.conv2d_same if padding == 'SAME' else TR.conv2d_valid
    h = conv(x, P['%s/vars/%s/weights' % (scope, name)], stride) + P['%s/vars/%s/bias' % (scope, name)]
    if norm == 'instance':
        h = instance_norm(h, P['%s/vars/%s/scale' % (scope, name)], P['%s/vars/%s/shift' % (scope, name)])
    elif norm in ('batch', 'renorm'):
        h = TR.batch_norm(h, bn_beta)
    return act(h) if act else h


def deconv2d(x, P, scope, name, act, norm=None, bn_beta=None):
    """hem/ops/layers.py:138-211 with the default output shape (2 x input, SAME)."""
    h = TR.conv2d_transpose_same(x, P['%s/vars/%s/weights' % (scope, name)], 2) + P['%s/vars/%s/bias' % (scope, name)]
    if norm == 'instance':
        h = instance_norm(h, P['%s/vars/%s/scale' % (scope, name)], P['%s/vars/%s/shift' % (scope, name)])
    elif norm in ('batch', 'renorm'):
        h = TR.batch_norm(h, bn_beta)
    return act(h) if act else h


def residual(x, P, scope, name, act, bn_betas=None):
    """hem/ops/layers.py:215-320 (stride 1, SAME): shortcut = convA(x) + bA;
    h = act([bn](shortcut)); h = [bn](convB(h) + bB); act(h + shortcut)."""
    h = TR.conv2d_same(x, P['%s/vars/%sA/weights' % (scope, name)], 1) + P['%s/vars/%sA/bias' % (scope, name)]   # :263-264
    shortcut = h                                                                                                 # :266
    if bn_betas is not None:
        h = TR.batch_norm(h, bn_betas[0])                                                                        # :273
    h = act(h) if act else h                                                                                     # :277
    h = TR.conv2d_same(h, P['%s/vars/%sB/weights' % (scope, name)], 1) + P['%s/vars/%sB/bias' % (scope, name)]   # :297-298
    if bn_betas is not None:
        h = TR.batch_norm(h, bn_betas[1])                                                                        # :303
    h = h + shortcut                                                                                             # :307
    return act(h) if act else h                                                                                  # :311
