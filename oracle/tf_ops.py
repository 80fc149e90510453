"""CPU restatement (NumPy) of the TensorFlow-1.x ops on the 3dgan training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`3dgan_amd/`) may import
this module; only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` do, and there only as the checker.

PARITY UNPINNED: the arithmetic of the reference lives in TensorFlow 1.x (tf.nn.conv2d,
tf.nn.conv2d_transpose, tf.contrib.layers.batch_norm, tf.gradients, tf.train.*Optimizer),
which is neither vendored under /root/reference nor installed here, and the reference's
own tests hold no vectors for these ops (only `hem/ops/test_losses.py:6-27`, rmse, which
`tests/test_oracle_known_answers.py` does check).  The semantics below are therefore the
published TF-1.x op contracts restated, pinned by analytic known
answers and by an independent torch-autograd implementation (`oracle/torch_ref.py`).

All functions are dtype-generic: pass float64 arrays for tight checks, float32 for a
reference-like run.  Layout is NHWC / HWIO throughout, as in gen-1 of the reference.

Reference call sites restated here:
  ops/layers.py:57      tf.matmul(x, W) + b                      -> dense
  ops/layers.py:101-102 tf.nn.conv2d(..., 'SAME') + bias_add      -> conv2d, bias_add
  ops/layers.py:140-143 tf.nn.conv2d_transpose(..., 'SAME')       -> conv2d_transpose
  ops/layers.py:58,103,144 tf.contrib.layers.batch_norm(h)        -> batch_norm_train
  ops/activations.py:28 tf.maximum(leak*x, x)                     -> lrelu
  models/gan.py:245,252,275  relu / tanh / sigmoid
  util.py:160-183       tf.train.{RMSProp,Adam,Momentum,GradientDescent}Optimizer
  hem/ops/losses.py:10-11  rmse
  hem/ops/images.py:53-70  rescale
"""
import math

import numpy as np

BN_EPS = 1e-3       # tf.contrib.layers.batch_norm default epsilon
BN_DECAY = 0.999    # default decay (moving stats are never updated by the reference, App. C-3)


# ----------------------------------------------------------------------------- padding
def same_pad(in_size, k, stride):
    """TF 'SAME': out = ceil(in/stride); extra padding goes after (bottom/right)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def valid_out(in_size, k, stride):
    return -(-(in_size - k + 1) // stride)


def _geometry(h, w, kh, kw, stride, padding):
    if padding == 'SAME':
        oh, pt, pb = same_pad(h, kh, stride)
        ow, pl, pr = same_pad(w, kw, stride)
    elif padding == 'VALID':
        oh, ow = valid_out(h, kh, stride), valid_out(w, kw, stride)
        pt = pb = pl = pr = 0
    else:
        raise ValueError(padding)
    return oh, ow, pt, pb, pl, pr


# ----------------------------------------------------------------------------- conv family
def conv2d(x, K, stride=1, padding='SAME'):
    """tf.nn.conv2d: cross-correlation, x [N,H,W,Cin], K [kh,kw,Cin,Cout] (ops/layers.py:101)."""
    n, h, w, cin = x.shape
    kh, kw, kcin, cout = K.shape
    assert kcin == cin
    oh, ow, pt, pb, pl, pr = _geometry(h, w, kh, kw, stride, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((n, oh, ow, cout), dtype=np.result_type(x, K))
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + stride * (oh - 1) + 1:stride, j:j + stride * (ow - 1) + 1:stride, :]
            y += patch @ K[i, j]
    return y


def conv2d_backprop_input(input_shape, K, dy, stride=1, padding='SAME'):
    """Gradient of conv2d w.r.t. its input; also *is* conv2d_transpose (App. A-2)."""
    n, h, w, cin = input_shape
    kh, kw, kcin, cout = K.shape
    assert kcin == cin and dy.shape[-1] == cout
    oh, ow, pt, pb, pl, pr = _geometry(h, w, kh, kw, stride, padding)
    assert dy.shape == (n, oh, ow, cout), (dy.shape, (n, oh, ow, cout))
    dxp = np.zeros((n, h + pt + pb, w + pl + pr, cin), dtype=np.result_type(dy, K))
    for i in range(kh):
        for j in range(kw):
            dxp[:, i:i + stride * (oh - 1) + 1:stride, j:j + stride * (ow - 1) + 1:stride, :] += dy @ K[i, j].T
    return dxp[:, pt:pt + h, pl:pl + w, :]


def conv2d_backprop_filter(x, filter_shape, dy, stride=1, padding='SAME'):
    """Gradient of conv2d w.r.t. its filter."""
    n, h, w, cin = x.shape
    kh, kw, kcin, cout = filter_shape
    oh, ow, pt, pb, pl, pr = _geometry(h, w, kh, kw, stride, padding)
    assert dy.shape == (n, oh, ow, cout)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dK = np.zeros(filter_shape, dtype=np.result_type(x, dy))
    dyf = dy.reshape(-1, cout)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + stride * (oh - 1) + 1:stride, j:j + stride * (ow - 1) + 1:stride, :]
            dK[i, j] = patch.reshape(-1, cin).T @ dyf
    return dK


def conv2d_transpose(x, K, output_shape, stride=2, padding='SAME'):
    """tf.nn.conv2d_transpose, filter [kh,kw,Cout,Cin] (ops/layers.py:135,142)."""
    return conv2d_backprop_input(tuple(output_shape), K, x, stride, padding)


def conv2d_transpose_backprop_input(K, dy, stride=2, padding='SAME'):
    """d(conv2d_transpose)/dx = forward conv of dy with the same filter."""
    return conv2d(dy, K, stride, padding)


def conv2d_transpose_backprop_filter(x, filter_shape, dy, stride=2, padding='SAME'):
    """d(conv2d_transpose)/dK: roles of input/output swap relative to conv2d."""
    return conv2d_backprop_filter(dy, filter_shape, x, stride, padding)


def bias_add(x, b):
    return x + b


def bias_grad(dy):
    return dy.reshape(-1, dy.shape[-1]).sum(axis=0)


def dense(x, W, b):
    """ops/layers.py:57"""
    return x @ W + b


# ----------------------------------------------------------------------------- batch norm
def batch_norm_train(x, beta, eps=BN_EPS):
    """tf.contrib.layers.batch_norm(h), all defaults, training mode (App. A-3):
    center=True, scale=False, biased variance over every axis but the last."""
    axes = tuple(range(x.ndim - 1))
    mean = x.mean(axis=axes)
    var = ((x - mean) ** 2).mean(axis=axes)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat + beta, (xhat, rstd, mean, var)


def batch_norm_train_backward(dy, cache):
    """Returns (dx, dbeta).  No gamma (scale=False)."""
    xhat, rstd, _, _ = cache
    axes = tuple(range(dy.ndim - 1))
    m = dy.size // dy.shape[-1]
    dbeta = dy.sum(axis=axes)
    dxhat_xhat = (dy * xhat).sum(axis=axes)
    dx = rstd * (dy - dbeta / m - xhat * dxhat_xhat / m)
    return dx, dbeta


# ----------------------------------------------------------------------------- activations
def lrelu(x, leak=0.2):
    """ops/activations.py:28  tf.maximum(leak*x, x)"""
    return np.maximum(leak * x, x)


def lrelu_grad_mask(x, leak=0.2):
    """d lrelu / dx.  TF's MaximumGrad routes the gradient to the first argument
    (leak*x) where leak*x >= x, i.e. at x <= 0 (for 0 <= leak < 1) the slope is `leak`."""
    return np.where(leak * x >= x, np.asarray(leak, dtype=x.dtype), np.asarray(1.0, dtype=x.dtype))


def relu(x):
    return np.maximum(x, 0)


def relu_grad_mask(x):
    return (x > 0).astype(x.dtype)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def tanh(x):
    return np.tanh(x)


ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4


def apply_act(x, act, leak=0.2):
    if act == ACT_NONE:
        return x
    if act == ACT_RELU:
        return relu(x)
    if act == ACT_LRELU:
        return lrelu(x, leak)
    if act == ACT_TANH:
        return tanh(x)
    if act == ACT_SIGMOID:
        return sigmoid(x)
    raise ValueError(act)


def act_backward(dy, pre, post, act, leak=0.2):
    """dL/dpre given dL/dpost."""
    if act == ACT_NONE:
        return dy
    if act == ACT_RELU:
        return dy * relu_grad_mask(pre)
    if act == ACT_LRELU:
        return dy * lrelu_grad_mask(pre, leak)
    if act == ACT_TANH:
        return dy * (1.0 - post * post)
    if act == ACT_SIGMOID:
        return dy * post * (1.0 - post)
    raise ValueError(act)


# ----------------------------------------------------------------------------- losses
def rmse(x, x_hat):
    """hem/ops/losses.py:10-11"""
    return np.sqrt(np.mean(np.square(x_hat - x)))


def rescale(x, orig=(-1, 1), new=(0, 1)):
    """hem/ops/images.py:68"""
    return (x - orig[0]) * (new[1] - new[0]) / (orig[1] - orig[0]) + new[0]


def sigmoid_cross_entropy_with_logits(logits, labels):
    """max(z,0) - z*l + log(1+exp(-|z|))  (App. A-7)"""
    z = logits
    return np.maximum(z, 0) - z * labels + np.log1p(np.exp(-np.abs(z)))


# ----------------------------------------------------------------------------- initialisers
def xavier_uniform(shape, rng, dtype=np.float32):
    """tf.contrib.layers.xavier_initializer() (uniform), used for weights AND biases
    (ops/layers.py:52-53,96-97,135-136; App. A-4)."""
    shape = tuple(int(s) for s in shape)
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


# ----------------------------------------------------------------------------- optimizers
class Adam:
    """tf.train.AdamOptimizer (App. A-5): eps outside the sqrt, lr_t folds both bias corrections."""

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def apply(self, params, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, g in grads.items():
            p = params[k]
            if k not in self.m:
                self.m[k] = np.zeros_like(p)
                self.v[k] = np.zeros_like(p)
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            params[k] = (p - lr_t * self.m[k] / (np.sqrt(self.v[k]) + self.eps)).astype(p.dtype)


class RMSProp:
    """tf.train.RMSPropOptimizer (App. A-5): rms slot starts at 1, eps=1e-10 inside the sqrt."""

    def __init__(self, lr=1e-3, decay=0.9, momentum=0.0, eps=1e-10, centered=False):
        self.lr, self.rho, self.mu, self.eps, self.centered = lr, decay, momentum, eps, centered
        self.rms, self.mom, self.mg = {}, {}, {}

    def apply(self, params, grads):
        for k, g in grads.items():
            p = params[k]
            if k not in self.rms:
                self.rms[k] = np.ones_like(p)
                self.mom[k] = np.zeros_like(p)
                self.mg[k] = np.zeros_like(p)
            self.rms[k] = self.rho * self.rms[k] + (1 - self.rho) * g * g
            denom = self.rms[k]
            if self.centered:
                self.mg[k] = self.rho * self.mg[k] + (1 - self.rho) * g
                denom = denom - self.mg[k] ** 2
            self.mom[k] = self.mu * self.mom[k] + self.lr * g / np.sqrt(denom + self.eps)
            params[k] = (p - self.mom[k]).astype(p.dtype)


class Momentum:
    """tf.train.MomentumOptimizer: acc = mu*acc + g; p -= lr*acc."""

    def __init__(self, lr=1e-3, momentum=0.0):
        self.lr, self.mu = lr, momentum
        self.acc = {}

    def apply(self, params, grads):
        for k, g in grads.items():
            p = params[k]
            if k not in self.acc:
                self.acc[k] = np.zeros_like(p)
            self.acc[k] = self.mu * self.acc[k] + g
            params[k] = (p - self.lr * self.acc[k]).astype(p.dtype)


class SGD:
    def __init__(self, lr=1e-3):
        self.lr = lr

    def apply(self, params, grads):
        for k, g in grads.items():
            params[k] = (params[k] - self.lr * g).astype(params[k].dtype)


class Adagrad:
    """tf.train.AdagradOptimizer: accumulator starts at 0.1; acc += g^2; p -= lr*g/sqrt(acc).  ProximalAdagrad with
    its default l1 = l2 = 0 reduces to the same update."""

    def __init__(self, lr=1e-3, initial_accumulator_value=0.1):
        self.lr, self.init = lr, initial_accumulator_value
        self.acc = {}

    def apply(self, params, grads):
        for k, g in grads.items():
            p = params[k]
            if k not in self.acc:
                self.acc[k] = np.full_like(p, self.init)
            self.acc[k] = self.acc[k] + g * g
            params[k] = (p - self.lr * g / np.sqrt(self.acc[k])).astype(p.dtype)


class Adadelta:
    """tf.train.AdadeltaOptimizer(lr, rho=0.95, epsilon=1e-8)."""

    def __init__(self, lr=1e-3, rho=0.95, eps=1e-8):
        self.lr, self.rho, self.eps = lr, rho, eps
        self.acc, self.acc_u = {}, {}

    def apply(self, params, grads):
        for k, g in grads.items():
            p = params[k]
            if k not in self.acc:
                self.acc[k] = np.zeros_like(p)
                self.acc_u[k] = np.zeros_like(p)
            self.acc[k] = self.rho * self.acc[k] + (1 - self.rho) * g * g
            u = np.sqrt(self.acc_u[k] + self.eps) / np.sqrt(self.acc[k] + self.eps) * g
            self.acc_u[k] = self.rho * self.acc_u[k] + (1 - self.rho) * u * u
            params[k] = (p - self.lr * u).astype(p.dtype)


class Ftrl:
    """tf.train.FtrlOptimizer(lr, learning_rate_power=-0.5, initial_accumulator_value=0.1, l1=0, l2=0)."""

    def __init__(self, lr=1e-3, initial_accumulator_value=0.1, l1=0.0, l2=0.0):
        self.lr, self.init, self.l1, self.l2 = lr, initial_accumulator_value, l1, l2
        self.acc, self.lin = {}, {}

    def apply(self, params, grads):
        for k, g in grads.items():
            p = params[k]
            if k not in self.acc:
                self.acc[k] = np.full_like(p, self.init)
                self.lin[k] = np.zeros_like(p)
            new_acc = self.acc[k] + g * g
            self.lin[k] = self.lin[k] + g - (np.sqrt(new_acc) - np.sqrt(self.acc[k])) / self.lr * p
            quad = np.sqrt(new_acc) / self.lr + 2 * self.l2
            lin = self.lin[k]
            params[k] = np.where(np.abs(lin) > self.l1, (np.sign(lin) * self.l1 - lin) / quad, 0).astype(p.dtype)
            self.acc[k] = new_acc


def init_optimizer(args):
    """util.py:150-183.  'pgd' returns None in the reference (missing return, :171-172)."""
    o = args.optimizer
    if o == 'rmsprop':
        return RMSProp(args.lr, decay=args.decay, momentum=args.momentum, centered=args.centered)
    if o == 'adam':
        return Adam(args.lr, args.beta1, args.beta2)
    if o == 'momentum':
        return Momentum(args.lr, args.momentum)
    if o == 'sgd':
        return SGD(args.lr)
    if o == 'adadelta':
        return Adadelta(args.lr)
    if o in ('adagrad', 'padagrad'):
        return Adagrad(args.lr)
    if o == 'ftrl':
        return Ftrl(args.lr)
    if o == 'pgd':
        return None
    raise ValueError('unknown optimizer %r' % o)


def average_gradients(tower_grads):
    """util.py:118-147: arithmetic mean over the tower axis, variable by variable."""
    out = {}
    for k in tower_grads[0]:
        out[k] = np.mean(np.stack([tg[k] for tg in tower_grads], axis=0), axis=0)
    return out
