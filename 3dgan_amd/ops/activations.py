"""Activation tokens accepted by the layer builders (reference: ops/activations.py:11-29 and
the tf.nn.* callables models/gan.py:245,252,275 pass as `activation=`).

In the reference these are graph-building callables; here each is a small callable object
carrying the code of the fused HIP epilogue (include/tdg.h TDG_ACT_*).  Calling one on a
symbolic tensor marks the producing layer's activation, so `activation(h)` in builder code
keeps working.
"""
from .._lib import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID


class Activation:
    def __init__(self, name, code, leak=0.0):
        self.name, self.code, self.leak = name, code, leak

    def __call__(self, x, leak=None, name=None):
        """Apply to a symbolic layer output (ops/layers.py:59,104,145 `activation(h)`)."""
        layer = x.producer
        if layer is None or layer.act is not None and layer.act.code != ACT_NONE:
            raise ValueError('activation %s: tensor has no producing layer to fuse into' % self.name)
        layer.act = self if leak is None else Activation(self.name, self.code, leak)
        return x

    def __repr__(self):
        return 'Activation(%s)' % self.name


def _lrelu(leak=0.2):
    return Activation('lrelu', ACT_LRELU, leak)


# ops/activations.py:11-29: tf.maximum(leak*x, x), leak = 0.2
lrelu = _lrelu(0.2)
relu = Activation('relu', ACT_RELU)          # tf.nn.relu
tanh = Activation('tanh', ACT_TANH)          # tf.tanh
sigmoid = Activation('sigmoid', ACT_SIGMOID)  # tf.nn.sigmoid
identity = Activation('none', ACT_NONE)


def selu(*_a, **_k):
    raise NotImplementedError('selu is unused by every model of the reference (SURVEY.md section 2 row 4)')
