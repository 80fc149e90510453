"""Layer builders with the signatures of the reference's `ops/layers.py:27-166`
(`dense`, `conv2d`, `deconv2d`, `flatten`, all `@add_arg_scope`) plus the small amount of
TF scaffolding the model files use around them (`arg_scope`, `variable_scope`, `reshape`,
`random_normal`).

In the reference each call adds TensorFlow graph nodes; here each call records a `LayerSpec`
into the `Net` of the enclosing variable scope.  The nets are later bound to HBM buffers and
executed by `3dgan_amd/engine.py`, every op being a HIP kernel behind include/tdg.h.  Variable
names follow the reference (`<scope>/vars/<name>/weights|bias`, ops/layers.py:15-24,51-53;
batch-norm betas `<scope>/BatchNorm[_i]/beta`).
"""
import functools
from contextlib import contextmanager

from . import activations as A

# ------------------------------------------------------------------------------ arg_scope
_arg_scopes = []


@contextmanager
def arg_scope(funcs, **kwargs):
    """tf.contrib.framework.arg_scope: default keyword arguments for the listed builders."""
    _arg_scopes.append({getattr(f, '_arg_scope_key', f): kwargs for f in funcs})
    try:
        yield
    finally:
        _arg_scopes.pop()


def add_arg_scope(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        merged = {}
        for scope in _arg_scopes:
            merged.update(scope.get(wrapper, {}))
        merged.update(kwargs)
        return func(*args, **merged)
    wrapper._arg_scope_key = wrapper
    return wrapper


def _resolve_activation(act, layer_name):
    """The reference passes arbitrary graph-building callables as `activation=` (e.g. `lambda x: hem.lrelu(x, leak=0.2)`,
    hem/models/paper_cgan.py:233, pix2pix.py:188).  Such a callable is traced once on a probe tensor; it must reduce to
    exactly one of the fused epilogue activations of ops/activations.py."""
    if act is None or isinstance(act, A.Activation):
        return act
    if not callable(act):
        raise TypeError('layer %s: activation %r is not callable' % (layer_name, act))

    class _Probe:
        act = None
    probe = Sym((None,), producer=_Probe())
    out = act(probe)
    if out is not probe or not isinstance(probe.producer.act, A.Activation):
        raise NotImplementedError('layer %s: activation %r does not reduce to one of relu / lrelu / tanh / sigmoid'
                                  % (layer_name, act))
    return probe.producer.act


# ------------------------------------------------------------------------------ nets / scopes
class LayerSpec:
    def __init__(self, kind, name, in_size, out_size, k=1, stride=1, use_bn=False, act=None,
                 in_shape=None, out_shape=None, padding='SAME', init='xavier', dropout=0):
        self.kind, self.name = kind, name
        self.dropout = dropout
        self.in_size, self.out_size, self.k, self.stride = in_size, out_size, k, stride
        self.use_bn, self.act = use_bn, _resolve_activation(act, name)
        self.in_shape, self.out_shape = in_shape, out_shape
        self.padding, self.init = padding, init

    def signature(self):
        return (self.kind, self.name, self.in_size, self.out_size, self.k, self.stride, self.use_bn,
                self.act.code if self.act else 0, self.in_shape, self.out_shape)

    @property
    def filter_shape(self):
        """Master variable shape (ops/layers.py:52,96,135)."""
        if self.kind == 'dense':
            return (self.in_size, self.out_size)
        if self.kind == 'conv2d':
            return (self.k, self.k, self.in_size, self.out_size)
        return (self.k, self.k, self.out_size, self.in_size)      # deconv2d: [k, k, Cout, Cin]


class Net:
    """All layers created under one variable scope ('generator', 'discriminator', ...).
    `passes[i]` is the layer list of the i-th call of the builder function; pass 0 creates
    the variables, later passes must reuse them (reuse=True), exactly like the reference."""

    def __init__(self, name):
        self.name = name
        self.passes = []
        self._cur = None

    @property
    def layers(self):
        return self.passes[0]

    def begin_pass(self):
        self._cur = []
        self.passes.append(self._cur)

    def add(self, spec, reuse):
        # a layer name seen again in the current pass means the builder function was called again
        if self._cur is None or any(l.name == spec.name for l in self._cur):
            self.begin_pass()
        idx = len(self._cur)
        if len(self.passes) > 1:
            if not reuse:
                raise ValueError('Variable %s/vars/%s/weights already exists (pass reuse=True)' % (self.name, spec.name))
            if idx >= len(self.layers) or self.layers[idx].signature() != spec.signature():
                raise ValueError('reuse=True but layer %s does not match the variables of the first pass' % spec.name)
        elif reuse:
            raise ValueError('Variable %s/vars/%s/weights does not exist (reuse=True on first use)' % (self.name, spec.name))
        self._cur.append(spec)
        return spec

    def bn_name(self, pass_idx, layer_idx):
        """contrib batch_norm uniquifies its default scope per call: BatchNorm, BatchNorm_1, ..."""
        per_pass = [i for i, l in enumerate(self.layers) if l.use_bn]
        i = pass_idx * len(per_pass) + per_pass.index(layer_idx)
        return '%s/BatchNorm%s/beta' % (self.name, '' if i == 0 else '_%d' % i)

    def var_name(self, layer, which):
        return '%s/vars/%s/%s' % (self.name, layer.name, which)


_nets = {}
_scope_stack = []


def reset_graph():
    _nets.clear()
    del _scope_stack[:]
    del _arg_scopes[:]


@contextmanager
def variable_scope(name):
    """tf.variable_scope(name): selects (or creates) the Net that collects the layers.
    Entering the same scope again starts a new pass over the same variables."""
    if _scope_stack:                                   # nested scopes join their names, as in TF
        name = _scope_stack[-1].name + '/' + name
    net = _nets.get(name)
    if net is None:
        net = _nets[name] = Net(name)
    _scope_stack.append(net)
    net._cur = None
    try:
        yield net
    finally:
        _scope_stack.pop()


def current_net():
    if not _scope_stack:
        raise RuntimeError('layer builders must be called inside variable_scope(...)')
    return _scope_stack[-1]


class Sym:
    """Symbolic tensor: static shape with a leading batch of None."""

    def __init__(self, shape, producer=None, source=None):
        self.shape, self.producer, self.source = tuple(shape), producer, source


def placeholder(shape, source='x'):
    return Sym(shape, source=source)


def random_normal(shape):
    """tf.random_normal (models/gan.py:246): drawn on-device per step (Philox), or injected."""
    return Sym((None,) + tuple(shape[1:]), source='random_normal')


def random_uniform(shape, minval=0.0, maxval=1.0):
    """tf.random_uniform (models/gan.py:224; hem/models/pix2pix.py:183,204,223): drawn on-device per step, or injected."""
    out = Sym((None,) + tuple(shape[1:]), source='random_uniform')
    out.minval, out.maxval = float(minval), float(maxval)
    return out


def reshape(x, shape):
    """tf.reshape on NHWC data: a reinterpretation, free when no channel padding is involved."""
    shape = tuple(None if s in (-1, None) else int(s) for s in shape)
    n_in = 1
    for s in x.shape[1:]:
        n_in *= s
    n_out = 1
    for s in shape[1:]:
        n_out *= s
    if n_in % n_out != 0:
        raise ValueError('cannot reshape %s to %s' % (x.shape, shape))
    out = Sym(shape, producer=x.producer, source=x.source)
    out.rows_per_image = getattr(x, 'rows_per_image', 1) * (n_in // n_out)
    return out


@add_arg_scope
def flatten(x, name=None):
    """ops/layers.py:152-166."""
    n = 1
    for s in x.shape[1:]:
        n *= s
    return reshape(x, [-1, n])


def _reject_unsupported(name, dropout, renorm, instance_norm):
    if dropout:
        raise NotImplementedError('layer %s: dropout > 0 is not available in this build (SURVEY.md section 2 row 12)' % name)
    if renorm or instance_norm:
        raise NotImplementedError('layer %s: batch-renorm / instance-norm are out of scope (thesis samplers only)' % name)


def concat(xs, axis=-1):
    """tf.concat along channels (axis=1 in the reference's NCHW == last axis in NHWC).  Executed as a
    zero-copy concat: producers write their channel window of one buffer (models/pix2pix.py)."""
    base = xs[0].shape[:-1]
    for t in xs:
        if t.shape[:-1] != base:
            raise ValueError('concat: spatial shapes differ: %s' % [t.shape for t in xs])
    out = Sym(base + (sum(t.shape[-1] for t in xs),), producer=None, source='concat')
    out.parts = list(xs)
    return out


def _same(in_size, k, stride):
    return -(-in_size // stride)


@add_arg_scope
def dense(x, input_size, output_size, init='xavier', use_batch_norm=False, activation=None, reuse=False, name=None):
    """ops/layers.py:27-62: h = x W + b [+ batch_norm] [+ activation]."""
    if x.shape[-1] != input_size or len(x.shape) != 2:
        raise ValueError('dense %s: input shape %s does not end in %d' % (name, x.shape, input_size))
    spec = LayerSpec('dense', name, input_size, output_size, use_bn=use_batch_norm, act=activation,
                     in_shape=(1, 1, input_size), out_shape=(1, 1, output_size), init=init)
    current_net().add(spec, reuse)
    out = Sym((None, output_size), producer=spec)
    out.rows_per_image = getattr(x, 'rows_per_image', 1)
    return out


@add_arg_scope
def conv2d(x, input_size, output_size, filter_size=3, stride=1, init='xavier', use_batch_norm=False,
           activation=None, reuse=False, name=None, padding='SAME', dropout=0, use_batch_renorm=False,
           use_instance_norm=False):
    """ops/layers.py:66-107 (gen-2: hem/ops/layers.py:70-135): tf.nn.conv2d + bias [+ batch_norm] [+ activation]."""
    _reject_unsupported(name, dropout, use_batch_renorm, use_instance_norm)
    _, h, w, c = x.shape
    if c != input_size:
        raise ValueError('conv2d %s: input has %d channels, expected %d' % (name, c, input_size))
    if padding == 'SAME':
        oh, ow = _same(h, filter_size, stride), _same(w, filter_size, stride)
    else:
        oh, ow = -(-(h - filter_size + 1) // stride), -(-(w - filter_size + 1) // stride)
    spec = LayerSpec('conv2d', name, input_size, output_size, filter_size, stride, use_batch_norm, activation,
                     (h, w, c), (oh, ow, output_size), padding, init)
    current_net().add(spec, reuse)
    return Sym((None, oh, ow, output_size), producer=spec)


@add_arg_scope
def deconv2d(x, input_size, output_size, filter_size=3, stride=2, init='xavier', use_batch_norm=False,
             activation=None, reuse=False, name=None, output_shape=None, dropout=0, use_batch_renorm=False,
             use_instance_norm=False, padding='SAME'):
    """ops/layers.py:111-148 (gen-2: hem/ops/layers.py:138-211): tf.nn.conv2d_transpose; output = 2 x input unless an
    explicit gen-2 `output_shape` = (N, C, H, W) is given (hem/ops/layers.py:185-187, used with padding='VALID' by
    hem/models/paper_cgan.py:237-241).  As in TF the shape must be one the forward conv maps back onto the input."""
    _reject_unsupported(name, 0, use_batch_renorm, use_instance_norm)        # dropout: recorded; the executor decides
    _, h, w, c = x.shape
    if c != input_size:
        raise ValueError('deconv2d %s: input has %d channels, expected %d' % (name, c, input_size))
    if padding not in ('SAME', 'VALID'):
        raise ValueError('deconv2d %s: padding %r' % (name, padding))
    if output_shape is None:
        oh, ow = h * 2, w * 2
    else:
        if len(output_shape) != 4 or int(output_shape[1]) != output_size:
            raise ValueError('deconv2d %s: output_shape %r is not (N, %d, H, W)' % (name, tuple(output_shape), output_size))
        oh, ow = int(output_shape[2]), int(output_shape[3])
    back = (lambda n: _same(n, filter_size, stride)) if padding == 'SAME' else \
        (lambda n: -(-(n - filter_size + 1) // stride))
    if (back(oh), back(ow)) != (h, w):
        raise ValueError('deconv2d %s: a %s k%d s%d conv of a %dx%d output does not give the %dx%d input'
                         % (name, padding, filter_size, stride, oh, ow, h, w))
    spec = LayerSpec('deconv2d', name, input_size, output_size, filter_size, stride, use_batch_norm, activation,
                     (h, w, c), (oh, ow, output_size), padding, init, dropout=dropout)
    current_net().add(spec, reuse)
    return Sym((None, oh, ow, output_size), producer=spec)
