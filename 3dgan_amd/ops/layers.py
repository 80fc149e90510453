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
                 in_shape=None, out_shape=None, padding='SAME', init='xavier', dropout=0, use_in=False):
        self.kind, self.name = kind, name
        self.dropout = dropout
        self.use_in = bool(use_in)                   # instance norm (hem/ops/images.py:73-89) between the conv and the activation
        self.in_size, self.out_size, self.k, self.stride = in_size, out_size, k, stride
        self.use_bn, self.act = use_bn, _resolve_activation(act, name)
        self.in_shape, self.out_shape = in_shape, out_shape
        self.padding, self.init = padding, init

    def signature(self):
        return (self.kind, self.name, self.in_size, self.out_size, self.k, self.stride, self.use_bn,
                self.act.code if self.act else 0, self.in_shape, self.out_shape, self.use_in)

    @property
    def normed(self):
        """The layer's output passes through a normalisation before its activation (batch norm or instance norm)."""
        return self.use_bn or self.use_in

    @property
    def n_bn(self):
        """Batch-norm calls (= `BatchNorm[_i]` scopes) this layer makes: a residual block normalises both of its convs."""
        return (2 if self.kind == 'residual' else 1) if self.use_bn else 0

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

    def bn_name(self, pass_idx, layer_idx, which=0):
        """contrib batch_norm uniquifies its default scope per call: BatchNorm, BatchNorm_1, ...  (`which`: the second
        batch norm of a residual block)."""
        counts = [l.n_bn for l in self.layers]
        i = pass_idx * sum(counts) + sum(counts[:layer_idx]) + which
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


def _reject_unsupported(name, dropout, renorm=False, instance_norm=False):
    if dropout:
        raise NotImplementedError('layer %s: dropout > 0 is not available in this build (SURVEY.md section 2 row 12)' % name)


def _norm_flags(name, use_batch_norm, use_batch_renorm, use_instance_norm):
    """(use_bn, use_in) of the gen-2 builders (hem/ops/layers.py:123-124,200-201).

    `use_batch_renorm=True` is `tf.contrib.layers.batch_norm(renorm=True)` in training mode.  Its correction terms are
    r = stddev / mixed_renorm_stddev and d = (mean - mixed_renorm_mean) / mixed_renorm_stddev, where the "mixed" moments
    zero-debias the renorm moving averages: (renorm_x + (1 - renorm_x_weight) * batch_x).  Those averages and their weights
    start at ZERO and are only advanced by the layer's update ops, which the reference never runs (SURVEY.md App. C-3:
    the same `control_dependencies` defect that disables weight clipping) -- so the mixed moments ARE the batch moments,
    r = 1 and d = 0 exactly, and batch renorm is batch norm: executed as such."""
    if use_instance_norm and (use_batch_norm or use_batch_renorm):
        raise NotImplementedError('layer %s: instance norm followed by batch norm is not built (no reference model does it)' % name)
    return bool(use_batch_norm or use_batch_renorm), bool(use_instance_norm)


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
    _reject_unsupported(name, dropout)
    use_batch_norm, use_in = _norm_flags(name, use_batch_norm, use_batch_renorm, use_instance_norm)
    _, h, w, c = x.shape
    if c != input_size:
        raise ValueError('conv2d %s: input has %d channels, expected %d' % (name, c, input_size))
    if padding == 'SAME':
        oh, ow = _same(h, filter_size, stride), _same(w, filter_size, stride)
    else:
        oh, ow = -(-(h - filter_size + 1) // stride), -(-(w - filter_size + 1) // stride)
    spec = LayerSpec('conv2d', name, input_size, output_size, filter_size, stride, use_batch_norm, activation,
                     (h, w, c), (oh, ow, output_size), padding, init, use_in=use_in)
    current_net().add(spec, reuse)
    return Sym((None, oh, ow, output_size), producer=spec)


@add_arg_scope
def residual(x, input_size, output_size, filter_size=3, stride=1, init='xavier', use_batch_norm=False, use_batch_renorm=False,
             use_instance_norm=False, activation=None, reuse=False, dropout=0, padding='SAME', name=None):
    """hem/ops/layers.py:215-320: a two-conv residual block.  conv A (`<name>A`) + bias -> `shortcut`; [batch norm],
    activation; conv B (`<name>B`, output_size -> output_size) + bias, [batch norm]; + shortcut; activation.
    Both convs use `stride`, so anything but stride 1 leaves the sum without matching shapes (TF raises there too)."""
    _reject_unsupported(name, dropout)
    if use_instance_norm:
        # instance_norm_op(h, reuse=reuse, name=name) is called twice with the same variable names: with reuse=False the
        # second get_variable raises in TF ("Variable ... already exists")
        raise NotImplementedError('residual %s: use_instance_norm cannot be built in the reference either (duplicate variables)' % name)
    use_bn, _ = _norm_flags(name, use_batch_norm, use_batch_renorm, False)
    _, h, w, c = x.shape
    if c != input_size:
        raise ValueError('residual %s: input has %d channels, expected %d' % (name, c, input_size))
    if stride != 1 or padding != 'SAME':
        raise ValueError('residual %s: the shortcut sum needs stride 1 and SAME padding (both convs use the same stride)' % name)
    spec = LayerSpec('residual', name, input_size, output_size, filter_size, 1, use_bn, activation,
                     (h, w, c), (h, w, output_size), padding, init)
    current_net().add(spec, reuse)
    return Sym((None, h, w, output_size), producer=spec)


@add_arg_scope
def deconv2d(x, input_size, output_size, filter_size=3, stride=2, init='xavier', use_batch_norm=False,
             activation=None, reuse=False, name=None, output_shape=None, dropout=0, use_batch_renorm=False,
             use_instance_norm=False, padding='SAME'):
    """ops/layers.py:111-148 (gen-2: hem/ops/layers.py:138-211): tf.nn.conv2d_transpose; output = 2 x input unless an
    explicit gen-2 `output_shape` = (N, C, H, W) is given (hem/ops/layers.py:185-187, used with padding='VALID' by
    hem/models/paper_cgan.py:237-241).  As in TF the shape must be one the forward conv maps back onto the input."""
    use_batch_norm, use_in = _norm_flags(name, use_batch_norm, use_batch_renorm, use_instance_norm)     # (dropout: recorded; the executor decides)
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
                     (h, w, c), (oh, ow, output_size), padding, init, dropout=dropout, use_in=use_in)
    current_net().add(spec, reuse)
    return Sym((None, oh, ow, output_size), producer=spec)
