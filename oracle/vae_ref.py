"""CPU restatement of the reference's conv VAE (`models/vae.py:25-151`): NumPy forward + hand-derived
backward, and an independent torch-autograd statement that pins it.

TEST INFRASTRUCTURE ONLY (see oracle/tf_ops.py).  PARITY UNPINNED against TensorFlow.

Effective semantics reproduced: only `decoder_loss` (the summed binary
cross-entropy, models/vae.py:76-77) is differentiated (`opt.compute_gradients(d_loss)`, :41); the KL
term and total are reported only; z_stddev is an unconstrained linear output (:125-128); the
second decoder pass on `samples` (:37) feeds summaries only and is not evaluated here; inputs stay
in [0,1] (no rescale); the encoder is hard-wired to 64x64 inputs through `32*4*4` (:125).
"""
import numpy as np
import torch

from . import tf_ops as T
from . import torch_ref as TR

ENC = [('c1', 3, 64, 5, 2), ('c2', 64, 128, 5, 2), ('c3', 128, 256, 5, 2), ('c4', 256, 256, 5, 2),
       ('c5', 256, 96, 1, 1), ('c6', 96, 32, 1, 1)]                       # models/vae.py:104-109, BN + lrelu
DEC_CONV = [('c1', 32, 96, 1, 1), ('c2', 96, 256, 1, 1)]                  # :145-146, relu, no BN
DEC_DECONV = [('dc1', 256, 256), ('dc2', 256, 128), ('dc3', 128, 64), ('dc4', 64, 3)]   # :147-150


def bn_name(i):
    return 'encoder/BatchNorm/beta' if i == 0 else 'encoder/BatchNorm_%d/beta' % i


def param_shapes(L, cin=3):
    sh = {}
    for i, (n, ci, co, k, s) in enumerate(ENC):
        ci = cin if i == 0 else ci
        sh['encoder/vars/%s/weights' % n] = (k, k, ci, co)
        sh['encoder/vars/%s/bias' % n] = (co,)
        sh[bn_name(i)] = (co,)
    for n in ('d1', 'd2'):
        sh['latent/vars/%s/weights' % n] = (512, L)
        sh['latent/vars/%s/bias' % n] = (L,)
    sh['decoder/vars/d1/weights'] = (L, 512)
    sh['decoder/vars/d1/bias'] = (512,)
    for n, ci, co, k, s in DEC_CONV:
        sh['decoder/vars/%s/weights' % n] = (k, k, ci, co)
        sh['decoder/vars/%s/bias' % n] = (co,)
    for n, ci, co in DEC_DECONV:
        co = cin if n == 'dc4' else co
        sh['decoder/vars/%s/weights' % n] = (5, 5, co, ci)
        sh['decoder/vars/%s/bias' % n] = (co,)
    return sh


def init_params(L, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return {k: (np.zeros(s, dtype) if k.endswith('/beta') else T.xavier_uniform(s, rng, dtype))
            for k, s in param_shapes(L).items()}


# ------------------------------------------------------------------------------ NumPy, hand backward
def forward(P, x, eps):
    c = {'x': x}
    h = x
    for i, (n, _, _, k, s) in enumerate(ENC):
        c['ein%d' % i] = h
        u = T.conv2d(h, P['encoder/vars/%s/weights' % n], s) + P['encoder/vars/%s/bias' % n]
        pre, c['ebn%d' % i] = T.batch_norm_train(u, P[bn_name(i)])
        c['epre%d' % i] = pre
        h = T.lrelu(pre)
    flat = h.reshape(h.shape[0], -1)
    c['flat'], c['eshape'] = flat, h.shape
    mean = flat @ P['latent/vars/d1/weights'] + P['latent/vars/d1/bias']
    std = flat @ P['latent/vars/d2/weights'] + P['latent/vars/d2/bias']
    z = mean + std * eps
    c.update(mean=mean, std=std, eps=eps, z=z)
    h = T.relu(z @ P['decoder/vars/d1/weights'] + P['decoder/vars/d1/bias'])
    c['dd1'] = h
    h = h.reshape(-1, 4, 4, 32)
    for i, (n, _, _, k, s) in enumerate(DEC_CONV):
        c['dcin%d' % i] = h
        h = T.relu(T.conv2d(h, P['decoder/vars/%s/weights' % n], s) + P['decoder/vars/%s/bias' % n])
        c['dcout%d' % i] = h
    for i, (n, _, _) in enumerate(DEC_DECONV):
        c['tin%d' % i] = h
        K = P['decoder/vars/%s/weights' % n]
        b, hh, ww, _ = h.shape
        u = T.conv2d_transpose(h, K, (b, hh * 2, ww * 2, K.shape[2]), 2) + P['decoder/vars/%s/bias' % n]
        h = T.sigmoid(u) if n == 'dc4' else T.relu(u)
        c['tout%d' % i] = h
    c['d'] = h
    d_loss = -np.sum(x * np.log(1e-8 + h) + (1 - x) * np.log(1e-8 + (1 - h)))                  # :76-77
    l_loss = 0.5 * np.sum(mean ** 2 + std ** 2 - np.log(1e-8 + std ** 2) - 1)                  # :80-81
    return {'decoder_loss': d_loss, 'latent_loss': l_loss, 'total_loss': d_loss + l_loss}, c


def backward(P, c, full_elbo=False):
    """Gradients of decoder_loss w.r.t. every trainable variable (models/vae.py:41)."""
    g = {}
    x, d = c['x'], c['d']
    dh = -(x / (1e-8 + d) - (1 - x) / (1e-8 + (1 - d)))
    for i in (3, 2, 1, 0):
        n = DEC_DECONV[i][0]
        out = c['tout%d' % i]
        du = dh * out * (1 - out) if n == 'dc4' else dh * (out > 0)
        K = P['decoder/vars/%s/weights' % n]
        g['decoder/vars/%s/bias' % n] = T.bias_grad(du)
        g['decoder/vars/%s/weights' % n] = T.conv2d_transpose_backprop_filter(c['tin%d' % i], K.shape, du, 2)
        dh = T.conv2d_transpose_backprop_input(K, du, 2)
    for i in (1, 0):
        n, _, _, k, s = DEC_CONV[i]
        du = dh * (c['dcout%d' % i] > 0)
        K = P['decoder/vars/%s/weights' % n]
        g['decoder/vars/%s/bias' % n] = T.bias_grad(du)
        g['decoder/vars/%s/weights' % n] = T.conv2d_backprop_filter(c['dcin%d' % i], K.shape, du, s)
        dh = T.conv2d_backprop_input(c['dcin%d' % i].shape, K, du, s)
    du = dh.reshape(dh.shape[0], -1) * (c['dd1'] > 0)
    g['decoder/vars/d1/bias'] = du.sum(0)
    g['decoder/vars/d1/weights'] = c['z'].T @ du
    dz = du @ P['decoder/vars/d1/weights'].T
    dmean, dstd = dz, dz * c['eps']
    if full_elbo:                                     # opt-in: + d latent_loss / d (mean, stddev)
        dmean = dmean + c['mean']
        dstd = dstd + c['std'] - c['std'] / (1e-8 + c['std'] ** 2)
    g['latent/vars/d1/weights'], g['latent/vars/d1/bias'] = c['flat'].T @ dmean, dmean.sum(0)
    g['latent/vars/d2/weights'], g['latent/vars/d2/bias'] = c['flat'].T @ dstd, dstd.sum(0)
    dh = (dmean @ P['latent/vars/d1/weights'].T + dstd @ P['latent/vars/d2/weights'].T).reshape(c['eshape'])
    for i in range(len(ENC) - 1, -1, -1):
        n, _, _, k, s = ENC[i]
        dpre = dh * T.lrelu_grad_mask(c['epre%d' % i])
        du, g[bn_name(i)] = T.batch_norm_train_backward(dpre, c['ebn%d' % i])
        K = P['encoder/vars/%s/weights' % n]
        g['encoder/vars/%s/bias' % n] = T.bias_grad(du)
        g['encoder/vars/%s/weights' % n] = T.conv2d_backprop_filter(c['ein%d' % i], K.shape, du, s)
        if i > 0:
            dh = T.conv2d_backprop_input(c['ein%d' % i].shape, K, du, s)
    return g


class VaeTrainer:
    """default_training (util.py:22-28): one sess.run of [train_op, losses] per call."""

    def __init__(self, P, args):
        self.P, self.opt = P, T.init_optimizer(args)

    def train_func(self, x, eps):
        losses, c = forward(self.P, x, eps)
        self.opt.apply(self.P, backward(self.P, c))
        return {k: float(v) for k, v in losses.items()}


# ------------------------------------------------------------------------------ torch autograd (independent)
def torch_losses(P, x, eps):
    h = x
    for i, (n, _, _, k, s) in enumerate(ENC):
        h = TR.conv2d_same(h, P['encoder/vars/%s/weights' % n], s) + P['encoder/vars/%s/bias' % n]
        h = TR.lrelu(TR.batch_norm(h, P[bn_name(i)]))
    flat = h.reshape(h.shape[0], -1)
    mean = flat @ P['latent/vars/d1/weights'] + P['latent/vars/d1/bias']
    std = flat @ P['latent/vars/d2/weights'] + P['latent/vars/d2/bias']
    z = mean + std * eps
    h = torch.relu(z @ P['decoder/vars/d1/weights'] + P['decoder/vars/d1/bias']).reshape(-1, 4, 4, 32)
    for n, _, _, k, s in DEC_CONV:
        h = torch.relu(TR.conv2d_same(h, P['decoder/vars/%s/weights' % n], s) + P['decoder/vars/%s/bias' % n])
    for n, _, _ in DEC_DECONV:
        u = TR.conv2d_transpose_same(h, P['decoder/vars/%s/weights' % n]) + P['decoder/vars/%s/bias' % n]
        h = torch.sigmoid(u) if n == 'dc4' else torch.relu(u)
    d_loss = -torch.sum(x * torch.log(1e-8 + h) + (1 - x) * torch.log(1e-8 + (1 - h)))
    l_loss = 0.5 * torch.sum(mean ** 2 + std ** 2 - torch.log(1e-8 + std ** 2) - 1)
    return d_loss, l_loss
