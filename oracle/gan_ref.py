"""CPU restatement (NumPy, hand-derived backward) of `models/gan.py` of the reference:
GAN / WGAN / IWGAN generator, discriminator, losses, gradient penalty (incl. the
second-order term) and the three train-step policies.

TEST INFRASTRUCTURE ONLY (see oracle/tf_ops.py header).  PARITY UNPINNED: TensorFlow is
not available, the reference holds no vectors for this path; this file is pinned by
`oracle/torch_ref.py` (independent torch-autograd statement, create_graph=True for the
penalty) and the analytic tests in tests/test_oracle_*.py.

Follows (all relative to /root/reference):
  models/gan.py:39-91    gan(x, args): flatten, 2*(x-0.5), per-tower G / D(real) / D(fake)
  models/gan.py:110-175  _train_gan / _train_wgan / _train_iwgan step policies
  models/gan.py:178-211  losses
  models/gan.py:214-231  gradient_penalty  (ONE norm over the whole batch tensor, :229)
  models/gan.py:234-254  generator
  models/gan.py:257-287  discriminator
  ops/layers.py:27-148   dense / conv2d / deconv2d (bias + optional batch_norm + activation)

Effective semantics reproduced: no weight clipping, no BN moving-average
updates (C-3); whole-batch gradient-penalty norm (C-4); D has no BN for iwgan; D(real) and
D(fake) own separate BN betas in gan/wgan (App. A-3); the reshape to [-1, 64L] before fc2
(C-2), which yields (H/8*W/8*4L)/(64L) rows per image.
"""
from types import SimpleNamespace

import numpy as np

from . import tf_ops as T


def make_cfg(model='iwgan', image_shape=(32, 32, 3), latent_size=200, batch_size=8):
    h, w, c = image_shape
    assert h % 16 == 0 and w % 16 == 0, 'generator doubles 4x from an s0 = H/16 base (App. C-1)'
    L = latent_size
    cfg = SimpleNamespace(model=model, H=h, W=w, C=c, L=L, B=batch_size,
                          s0h=h // 16, s0w=w // 16,
                          d_bn=(model != 'iwgan'),                       # models/gan.py:274
                          d_sigmoid=(model == 'gan'))                    # models/gan.py:275
    # D conv stack: 3 stride-2 SAME convs -> ceil(H/8) x ceil(W/8) x 4L, reshaped to [-1, 4*4*4L]
    fh, fw = -(-h // 8), -(-w // 8)
    feat = fh * fw * 4 * L
    assert feat % (64 * L) == 0
    cfg.d_rows_per_image = feat // (64 * L)
    cfg.fc2_in = 64 * L
    return cfg


# --------------------------------------------------------------------------- parameters
G_BN_LAYERS = ['fc1', 'dc1', 'dc2', 'dc3']


def g_bn_name(i):
    return 'generator/BatchNorm/beta' if i == 0 else 'generator/BatchNorm_%d/beta' % i


def d_bn_name(pass_idx, layer_idx):
    """pass 0 = D(real), 1 = D(fake); layer 0 = c2, 1 = c3.  contrib batch_norm
    uniquifies its default scope per call site: BatchNorm, BatchNorm_1, ..."""
    i = pass_idx * 2 + layer_idx
    return 'discriminator/BatchNorm/beta' if i == 0 else 'discriminator/BatchNorm_%d/beta' % i


def param_shapes(cfg):
    L, C = cfg.L, cfg.C
    sh = {}
    g = 'generator/vars/'
    sh[g + 'fc1/weights'] = (L, cfg.s0h * cfg.s0w * 4 * L)
    sh[g + 'fc1/bias'] = (cfg.s0h * cfg.s0w * 4 * L,)
    # deconv filters are [k, k, Cout, Cin] (ops/layers.py:135)
    for name, cin, cout in [('dc1', 4 * L, 2 * L), ('dc2', 2 * L, L), ('dc3', L, L // 2), ('dc4', L // 2, C)]:
        sh[g + name + '/weights'] = (5, 5, cout, cin)
        sh[g + name + '/bias'] = (cout,)
    for i, n in enumerate([cfg.s0h * cfg.s0w * 4 * L, 2 * L, L, L // 2]):
        sh[g_bn_name(i)] = (n,)
    d = 'discriminator/vars/'
    for name, cin, cout in [('c1', C, L), ('c2', L, 2 * L), ('c3', 2 * L, 4 * L)]:
        sh[d + name + '/weights'] = (5, 5, cin, cout)
        sh[d + name + '/bias'] = (cout,)
    sh[d + 'fc2/weights'] = (cfg.fc2_in, 1)
    sh[d + 'fc2/bias'] = (1,)
    if cfg.d_bn:
        for p in range(2):
            for li, n in enumerate([2 * L, 4 * L]):
                sh[d_bn_name(p, li)] = (n,)
    return sh


def init_params(cfg, seed=0, dtype=np.float32):
    """xavier-uniform for weights and biases, zeros for BN beta (App. A-3/A-4)."""
    rng = np.random.default_rng(seed)
    P = {}
    for k, s in param_shapes(cfg).items():
        if k.endswith('/beta'):
            P[k] = np.zeros(s, dtype=dtype)
        else:
            P[k] = T.xavier_uniform(s, rng, dtype)
    return P


# Test hook ("kink-resolved" comparisons, tests/test_gpu_distributed.py, tests/test_oracle_towers.py): an object with
#   mask(tag, layer, pre, default, lo) -> derivative mask
# may flip the (l)relu derivative (between 1 and `lo`) of entries whose pre-activation lies within floating-point rounding
# of the kink.  `tag` names the pass ('real' / 'fake' / 'hat' of the critic, 'g' of the generator); the first-order backward
# and the penalty's tangent pass ask with the same (tag, layer), so a flip is applied consistently.  None: plain masks.
KINK = None


def _act_mask(tag, layer, pre, default, lo):
    return default if KINK is None else KINK.mask(tag, layer, pre, default, lo)


def split_params(P):
    g = {k: v for k, v in P.items() if k.startswith('generator/')}
    d = {k: v for k, v in P.items() if k.startswith('discriminator/')}
    return g, d


# --------------------------------------------------------------------------- generator
def g_forward(P, z, cfg):
    """models/gan.py:234-254 with the 32-native generalisation of SURVEY App. C-1."""
    L = cfg.L
    g = 'generator/vars/'
    cache = {'z': z}
    h = T.dense(z, P[g + 'fc1/weights'], P[g + 'fc1/bias'])
    h, cache['bn0'] = T.batch_norm_train(h, P[g_bn_name(0)])
    cache['pre0'] = h
    h = T.relu(h)
    h = h.reshape(-1, cfg.s0h, cfg.s0w, 4 * L)
    specs = [('dc1', 2 * L, True), ('dc2', L, True), ('dc3', L // 2, True), ('dc4', cfg.C, False)]
    for i, (name, cout, bn) in enumerate(specs, start=1):
        cache['in%d' % i] = h
        n, hh, ww, _ = h.shape
        y = T.conv2d_transpose(h, P[g + name + '/weights'], (n, hh * 2, ww * 2, cout), 2)
        y = T.bias_add(y, P[g + name + '/bias'])
        if bn:
            y, cache['bn%d' % i] = T.batch_norm_train(y, P[g_bn_name(i)])
            cache['pre%d' % i] = y
            h = T.relu(y)
        else:
            cache['pre%d' % i] = y
            h = T.tanh(y)
            cache['post%d' % i] = h
    return h.reshape(h.shape[0], -1), cache


def g_backward(P, cache, dg_flat, cfg):
    """dL/d(generator params) given dL/dg."""
    L = cfg.L
    g = 'generator/vars/'
    grads = {}
    n = dg_flat.shape[0]
    dh = dg_flat.reshape(n, cfg.H, cfg.W, cfg.C)
    specs = [('dc1', True), ('dc2', True), ('dc3', True), ('dc4', False)]
    for i in range(4, 0, -1):
        name, bn = specs[i - 1]
        if bn:
            dpre = dh * _act_mask('g', i, cache['pre%d' % i], T.relu_grad_mask(cache['pre%d' % i]), 0.0)
            dy, grads[g_bn_name(i)] = T.batch_norm_train_backward(dpre, cache['bn%d' % i])
        else:
            dy = dh * (1.0 - cache['post%d' % i] ** 2)
        grads[g + name + '/bias'] = T.bias_grad(dy)
        K = P[g + name + '/weights']
        grads[g + name + '/weights'] = T.conv2d_transpose_backprop_filter(cache['in%d' % i], K.shape, dy, 2)
        dh = T.conv2d_transpose_backprop_input(K, dy, 2)
    dh = dh.reshape(n, -1)
    dpre = dh * _act_mask('g', 0, cache['pre0'], T.relu_grad_mask(cache['pre0']), 0.0)
    dy, grads[g_bn_name(0)] = T.batch_norm_train_backward(dpre, cache['bn0'])
    grads[g + 'fc1/bias'] = dy.sum(axis=0)
    grads[g + 'fc1/weights'] = cache['z'].T @ dy
    return grads


# --------------------------------------------------------------------------- discriminator
def d_forward(P, x_flat, cfg, bn_pass=0, tag=None):
    """models/gan.py:257-287.  Returns (d [rows], cache).  `bn_pass` selects the beta set."""
    L = cfg.L
    d = 'discriminator/vars/'
    cache = {'tag': tag or ('real', 'fake')[bn_pass]}
    h = x_flat.reshape(-1, cfg.H, cfg.W, cfg.C)
    for i, name in enumerate(['c1', 'c2', 'c3']):
        cache['in%d' % i] = h
        y = T.bias_add(T.conv2d(h, P[d + name + '/weights'], 2), P[d + name + '/bias'])
        if cfg.d_bn and i > 0:
            y, cache['bn%d' % i] = T.batch_norm_train(y, P[d_bn_name(bn_pass, i - 1)])
        cache['pre%d' % i] = y
        h = T.lrelu(y)
    cache['feat_shape'] = h.shape
    f = h.reshape(-1, cfg.fc2_in)                                    # models/gan.py:284
    cache['feat'] = f
    o = T.dense(f, P[d + 'fc2/weights'], P[d + 'fc2/bias'])
    if cfg.d_sigmoid:
        cache['logit'] = o
        o = T.sigmoid(o)
    cache['out'] = o
    return o.reshape(-1), cache


def d_backward(P, cache, dd, cfg, bn_pass=0, want_params=True, want_dx=True):
    """Backprop dL/dd [rows] through D.  Returns (dx_flat or None, grads, deltas) where
    deltas[i] is dL/d(conv_i output incl. bias) -- reused by the penalty's second order."""
    d = 'discriminator/vars/'
    grads = {}
    do = dd.reshape(-1, 1)
    if cfg.d_sigmoid:
        o = cache['out']
        do = do * o * (1.0 - o)
    if want_params:
        grads[d + 'fc2/weights'] = cache['feat'].T @ do
        grads[d + 'fc2/bias'] = do.sum(axis=0)
    dh = (do @ P[d + 'fc2/weights'].T).reshape(cache['feat_shape'])
    deltas = {}
    dx = None
    for i in (2, 1, 0):
        name = ['c1', 'c2', 'c3'][i]
        dy = dh * _act_mask(cache['tag'], i, cache['pre%d' % i], T.lrelu_grad_mask(cache['pre%d' % i]), 0.2)
        if cfg.d_bn and i > 0:
            dy, dbeta = T.batch_norm_train_backward(dy, cache['bn%d' % i])
            if want_params:
                grads[d_bn_name(bn_pass, i - 1)] = dbeta
        deltas[i] = dy
        K = P[d + name + '/weights']
        if want_params:
            grads[d + name + '/bias'] = T.bias_grad(dy)
            grads[d + name + '/weights'] = T.conv2d_backprop_filter(cache['in%d' % i], K.shape, dy, 2)
        if i > 0 or want_dx:
            dh = T.conv2d_backprop_input(cache['in%d' % i].shape, K, dy, 2)
    if want_dx:
        dx = dh.reshape(dh.shape[0], -1)
    return dx, grads, deltas


def add_grads(a, b, scale=1.0):
    for k, v in b.items():
        a[k] = a[k] + scale * v if k in a else scale * v
    return a


# --------------------------------------------------------------------------- gradient penalty
def gradient_penalty(P, x, g, alpha, cfg, want_param_grads=True):
    """models/gan.py:214-231.  penalty = (sqrt(sum_ALL grad^2) - 1)^2 -- one scalar norm
    over the whole [B, HWC] tensor (C-4).  Returns (penalty, grads_of_penalty_wrt_D_params).

    Second order: D for iwgan is piecewise linear (lrelu, no BN), so with the
    masks M_i held constant  u . grad_xhat(sum D(xhat)) = < M3 W3 M2 W2 M1 W1 u , w4 >, whose
    parameter gradient is an ordinary backward pass of the *tangent* network fed with u:
    dW_i = bwd_filter(t_{i-1}, delta_i), t_i = M_i * conv_i(t_{i-1}) (no bias), with delta_i the
    first-order deltas.  Biases receive no gradient from the penalty."""
    assert not cfg.d_bn, 'gradient penalty is only defined for the BN-free iwgan critic'
    d = 'discriminator/vars/'
    xhat = x + alpha * (g - x)                                        # :225-226
    dhat, cache = d_forward(P, xhat, cfg, tag='hat')
    ones = np.ones_like(dhat)                                         # tf.gradients sums the outputs
    v, _, deltas = d_backward(P, cache, ones, cfg, want_params=False, want_dx=True)
    if getattr(cfg, 'gp_per_sample', False):                          # opt-in: one norm per image
        s = np.sqrt(np.sum(np.square(v), axis=1, keepdims=True))
        penalty = float(np.mean((s - 1.0) ** 2))
        u = (2.0 * (s - 1.0) / s / v.shape[0]) * v
    else:
        s = np.sqrt(np.sum(np.square(v)))                             # :229
        penalty = (s - 1.0) ** 2                                      # :230
        u = (2.0 * (s - 1.0) / s) * v                                 # d penalty / d v
    if not want_param_grads:
        return penalty, {}
    grads = {}
    t = u.reshape(-1, cfg.H, cfg.W, cfg.C)
    for i, name in enumerate(['c1', 'c2', 'c3']):
        K = P[d + name + '/weights']
        grads[d + name + '/weights'] = T.conv2d_backprop_filter(t, K.shape, deltas[i], 2)
        grads[d + name + '/bias'] = np.zeros_like(P[d + name + '/bias'])
        t = T.conv2d(t, K, 2) * _act_mask('hat', i, cache['pre%d' % i], T.lrelu_grad_mask(cache['pre%d' % i]), 0.2)
    tf_ = t.reshape(-1, cfg.fc2_in)
    grads[d + 'fc2/weights'] = tf_.sum(axis=0).reshape(-1, 1)
    grads[d + 'fc2/bias'] = np.zeros_like(P[d + 'fc2/bias'])
    return penalty, grads


# --------------------------------------------------------------------------- losses + grads
GP_LAMBDA = 10.0     # models/gan.py:199


def d_loss_and_grads(P, x, z, alpha, cfg, want_grads=True):
    """d_loss and its gradient w.r.t. the discriminator variables (models/gan.py:63,68)."""
    g, _ = g_forward(P, z, cfg)
    d_real, c_real = d_forward(P, x, cfg, 0)
    d_fake, c_fake = d_forward(P, g, cfg, 1)
    nr = d_real.size
    grads = {}
    if cfg.model == 'gan':                                             # :193-194
        loss = np.mean(-np.log(d_real + 1e-8) - np.log(1.0 - d_fake + 1e-8))
        dd_real = -1.0 / (d_real + 1e-8) / nr
        dd_fake = 1.0 / (1.0 - d_fake + 1e-8) / nr
    else:                                                              # :197, :204
        loss = np.mean(d_fake) - np.mean(d_real)
        dd_real = -np.ones_like(d_real) / nr
        dd_fake = np.ones_like(d_fake) / nr
    if want_grads:
        _, gr, _ = d_backward(P, c_real, dd_real, cfg, 0, True, False)
        _, gf, _ = d_backward(P, c_fake, dd_fake, cfg, 1, True, False)
        add_grads(grads, gr)
        add_grads(grads, gf)
    gp = 0.0
    if cfg.model == 'iwgan':
        gp, ggp = gradient_penalty(P, x, g, alpha, cfg, want_grads)
        loss = loss + GP_LAMBDA * gp
        if want_grads:
            add_grads(grads, ggp, GP_LAMBDA)
    return loss, grads, dict(g=g, d_real=d_real, d_fake=d_fake, gp=gp)


def g_loss_and_grads(P, z, cfg, want_grads=True):
    """g_loss and its gradient w.r.t. the generator variables (models/gan.py:63,67)."""
    g, gc = g_forward(P, z, cfg)
    d_fake, c_fake = d_forward(P, g, cfg, 1)
    n = d_fake.size
    if cfg.model == 'gan':
        loss = np.mean(-np.log(d_fake + 1e-8))
        dd = -1.0 / (d_fake + 1e-8) / n
    else:
        loss = -np.mean(d_fake)
        dd = -np.ones_like(d_fake) / n
    grads = {}
    if want_grads:
        dg, _, _ = d_backward(P, c_fake, dd, cfg, 1, False, True)
        grads = g_backward(P, gc, dg, cfg)
    return loss, grads, dict(g=g, d_fake=d_fake)


# --------------------------------------------------------------------------- train-step policies
class GanTrainer:
    """One replica ("tower") of models/gan.py's train_func.  Inputs that TF would draw from
    its unseeded RNG (z, alpha) and the batches are injected."""

    def __init__(self, P, cfg, args):
        self.P, self.cfg, self.args = P, cfg, args
        self.g_opt, self.d_opt = T.init_optimizer(args), T.init_optimizer(args)   # models/gan.py:46
        self.global_step = 0

    def rescale(self, x01):
        """models/gan.py:49-50: flatten, 2*(x-0.5)."""
        return 2.0 * (x01.reshape(x01.shape[0], -1) - 0.5)

    def d_step(self, x01, z, alpha=None):
        loss, grads, _ = d_loss_and_grads(self.P, self.rescale(x01), z, alpha, self.cfg)
        self.d_opt.apply(self.P, grads)
        self.global_step += 1
        return loss

    def g_step(self, x01, z, alpha=None):
        """Also evaluates the displayed losses on this batch (models/gan.py:172)."""
        gl, grads, _ = g_loss_and_grads(self.P, z, self.cfg)
        dl, _, _ = d_loss_and_grads(self.P, self.rescale(x01), z, alpha, self.cfg, want_grads=False)
        self.g_opt.apply(self.P, grads)
        self.global_step += 1
        return {'g_loss': float(gl), 'd_loss': float(dl)}

    def train_func(self, batches, zs, alphas=None):
        """models/gan.py:110-175.  wgan/iwgan: n_disc_train D steps then one G step, each on
        a fresh batch; gan: one batch, both updates from the same forward."""
        n = self.args.n_disc_train
        if self.cfg.model == 'gan':
            x = self.rescale(batches[0])
            gl, gg, _ = g_loss_and_grads(self.P, zs[0], self.cfg)
            dl, dg, _ = d_loss_and_grads(self.P, x, zs[0], None, self.cfg)
            self.d_opt.apply(self.P, dg)
            self.g_opt.apply(self.P, gg)
            self.global_step += 2
            return {'g_loss': float(gl), 'd_loss': float(dl)}
        for i in range(n):
            self.d_step(batches[i], zs[i], None if alphas is None else alphas[i])
        return self.g_step(batches[n], zs[n], None if alphas is None else alphas[n])
