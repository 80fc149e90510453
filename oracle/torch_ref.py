"""Independent CPU statement of the same path on torch autograd (no hand-derived backward):
used (a) to pin `oracle/gan_ref.py` / `oracle/tf_ops.py` in tests and (b) as the
multi-threaded `cpu_baseline` ("port") that bench.py times on the GPU box's host cores.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.  PARITY UNPINNED against
TensorFlow (see oracle/tf_ops.py header).

TF semantics are spelled out explicitly rather than borrowed from torch defaults:
asymmetric SAME padding via F.pad (App. A-1), conv2d_transpose as scatter + crop (A-2),
batch_norm without gamma, biased variance, eps 1e-3 (A-3), lrelu = max(leak*x, x) (A-6),
whole-batch gradient-penalty norm (models/gan.py:229) through autograd.grad(create_graph=True).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .tf_ops import same_pad, BN_EPS
from . import gan_ref


def conv2d_same(x, K, stride):
    """x NHWC, K HWIO."""
    n, h, w, c = x.shape
    kh, kw = K.shape[0], K.shape[1]
    _, pt, pb = same_pad(h, kh, stride)
    _, pl, pr = same_pad(w, kw, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, K.permute(3, 2, 0, 1).contiguous(), stride=stride)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose_same(x, K, stride=2):
    """x NHWC [N,h,w,Cin], K [kh,kw,Cout,Cin]; output [N, stride*h, stride*w, Cout]
    (ops/layers.py:140-142).  Full scatter, then crop the forward conv's SAME padding."""
    n, h, w, cin = x.shape
    kh, kw = K.shape[0], K.shape[1]
    H, W = h * stride, w * stride
    _, pt, pb = same_pad(H, kh, stride)
    _, pl, pr = same_pad(W, kw, stride)
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2).contiguous(), K.permute(3, 2, 0, 1).contiguous(), stride=stride)
    # full size = (h-1)*stride + kh ; padded forward input size = H + pt + pb >= that
    fh, fw = full.shape[2], full.shape[3]
    full = F.pad(full, (0, W + pl + pr - fw, 0, H + pt + pb - fh))
    y = full[:, :, pt:pt + H, pl:pl + W]
    return y.permute(0, 2, 3, 1)


def batch_norm(x, beta):
    axes = tuple(range(x.dim() - 1))
    mean = x.mean(dim=axes)
    var = ((x - mean) ** 2).mean(dim=axes)
    return (x - mean) * torch.rsqrt(var + BN_EPS) + beta


def conv2d_valid(x, K, stride):
    """tf.nn.conv2d padding='VALID' (SURVEY App. A-1): x NHWC, K HWIO, out = ceil((in - k + 1) / stride)."""
    y = F.conv2d(x.permute(0, 3, 1, 2), K.permute(3, 2, 0, 1).contiguous(), stride=stride)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose_valid(x, K, out_hw, stride=2):
    """tf.nn.conv2d_transpose padding='VALID' with an explicit output_shape (hem/ops/layers.py:185-193): the adjoint of
    the VALID conv [N,H,W,Cout] -> [N,h,w,Cin]; rows/cols of the output that no window of that conv reads stay zero."""
    H, W = out_hw
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2).contiguous(), K.permute(3, 2, 0, 1).contiguous(), stride=stride)
    fh, fw = full.shape[2], full.shape[3]
    assert 0 <= H - fh < stride and 0 <= W - fw < stride, 'output_shape is not consistent with the VALID geometry'
    return F.pad(full, (0, W - fw, 0, H - fh)).permute(0, 2, 3, 1)


def lrelu(x, leak=0.2):
    return torch.maximum(leak * x, x)


# Test hook (tests/test_gpu_headline_parity.py): FORCED derivative masks.  MASKS maps (tag, layer) -> bool array, True where
# the activation's derivative is 1 (else `lo`: the leak, or 0 for relu); tag 'g' = generator layers 0..3, 'real' / 'fake' /
# 'hat' = the critic's passes.  The activation's VALUE is untouched; only d act / d pre is the given mask instead of the
# sign of the oracle's own pre-activation -- they differ exactly where a pre-activation lies within the other side's
# rounding of the kink.  None (default): plain autograd.
MASKS = None


def _act(y, tag, layer, lo):
    val = torch.maximum(lo * y, y)
    if MASKS is None or (tag, layer) not in MASKS:
        return val
    m = torch.as_tensor(MASKS[(tag, layer)]).to(y.dtype).reshape(y.shape)
    lin = y * (m + lo * (1.0 - m))
    return lin + (val - lin).detach()


def generator(P, z, cfg):
    g = 'generator/vars/'
    L = cfg.L
    h = z @ P[g + 'fc1/weights'] + P[g + 'fc1/bias']
    h = _act(batch_norm(h, P[gan_ref.g_bn_name(0)]), 'g', 0, 0.0)
    h = h.reshape(-1, cfg.s0h, cfg.s0w, 4 * L)
    for i, name in enumerate(['dc1', 'dc2', 'dc3'], start=1):
        h = conv2d_transpose_same(h, P[g + name + '/weights']) + P[g + name + '/bias']
        h = _act(batch_norm(h, P[gan_ref.g_bn_name(i)]), 'g', i, 0.0)
    h = torch.tanh(conv2d_transpose_same(h, P[g + 'dc4/weights']) + P[g + 'dc4/bias'])
    return h.reshape(h.shape[0], -1)


def discriminator(P, x_flat, cfg, bn_pass=0, tag=None):
    d = 'discriminator/vars/'
    tag = tag or ('real', 'fake')[bn_pass]
    h = x_flat.reshape(-1, cfg.H, cfg.W, cfg.C)
    for i, name in enumerate(['c1', 'c2', 'c3']):
        h = conv2d_same(h, P[d + name + '/weights'], 2) + P[d + name + '/bias']
        if cfg.d_bn and i > 0:
            h = batch_norm(h, P[gan_ref.d_bn_name(bn_pass, i - 1)])
        h = _act(h, tag, i, 0.2)
    h = h.reshape(-1, cfg.fc2_in)
    o = h @ P[d + 'fc2/weights'] + P[d + 'fc2/bias']
    if cfg.d_sigmoid:
        o = torch.sigmoid(o)
    return o.reshape(-1)


def losses(P, x, z, alpha, cfg):
    """models/gan.py:178-231 (x already flattened and rescaled to [-1,1])."""
    g = generator(P, z, cfg)
    d_real = discriminator(P, x, cfg, 0)
    d_fake = discriminator(P, g, cfg, 1)
    if cfg.model == 'gan':
        g_loss = torch.mean(-torch.log(d_fake + 1e-8))
        d_loss = torch.mean(-torch.log(d_real + 1e-8) - torch.log(1 - d_fake + 1e-8))
    else:
        g_loss = -torch.mean(d_fake)
        d_loss = torch.mean(d_fake) - torch.mean(d_real)
        if cfg.model == 'iwgan':
            xhat = x + alpha * (g - x)
            d_hat = discriminator(P, xhat, cfg, 0, tag='hat')
            grad = torch.autograd.grad(d_hat.sum(), xhat, create_graph=True)[0]
            slopes = torch.sqrt(torch.sum(grad ** 2))
            d_loss = d_loss + gan_ref.GP_LAMBDA * (slopes - 1.0) ** 2
    return g_loss, d_loss


def to_torch(P, dtype=torch.float32):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in P.items()}


def grads_of(loss, P, prefix):
    keys = [k for k in P if k.startswith(prefix)]
    gs = torch.autograd.grad(loss, [P[k] for k in keys], allow_unused=True, retain_graph=True)
    return {k: (torch.zeros_like(P[k]) if g is None else g) for k, g in zip(keys, gs)}


# ----------------------------------------------------------------------------- optimizers (TF rules)
class TorchAdam:
    def __init__(self, lr, b1, b2, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m, self.v = {}, {}

    @torch.no_grad()
    def apply(self, P, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, g in grads.items():
            if k not in self.m:
                self.m[k] = torch.zeros_like(g)
                self.v[k] = torch.zeros_like(g)
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            P[k].sub_(lr_t * self.m[k] / (self.v[k].sqrt() + self.eps))


class TorchRMSProp:
    def __init__(self, lr, decay=0.9, momentum=0.0, eps=1e-10):
        self.lr, self.rho, self.mu, self.eps = lr, decay, momentum, eps
        self.rms, self.mom = {}, {}

    @torch.no_grad()
    def apply(self, P, grads):
        for k, g in grads.items():
            if k not in self.rms:
                self.rms[k] = torch.ones_like(g)
                self.mom[k] = torch.zeros_like(g)
            self.rms[k].mul_(self.rho).addcmul_(g, g, value=1 - self.rho)
            self.mom[k].mul_(self.mu).add_(self.lr * g / torch.sqrt(self.rms[k] + self.eps))
            P[k].sub_(self.mom[k])


def make_optimizer(args):
    if args.optimizer == 'adam':
        return TorchAdam(args.lr, args.beta1, args.beta2)
    if args.optimizer == 'rmsprop':
        return TorchRMSProp(args.lr, args.decay, args.momentum)
    raise NotImplementedError(args.optimizer)


class TorchGanTrainer:
    """Same step policy as gan_ref.GanTrainer, on autograd.  Used as the timed CPU baseline."""

    def __init__(self, P, cfg, args):
        self.P, self.cfg, self.args = P, cfg, args
        self.g_opt, self.d_opt = make_optimizer(args), make_optimizer(args)

    @staticmethod
    def rescale(x01):
        return 2.0 * (x01.reshape(x01.shape[0], -1) - 0.5)

    def d_step(self, x01, z, alpha):
        _, d_loss = losses(self.P, self.rescale(x01), z, alpha, self.cfg)
        self.d_opt.apply(self.P, grads_of(d_loss, self.P, 'discriminator/'))
        return float(d_loss.detach())

    def g_step(self, x01, z, alpha):
        g_loss, d_loss = losses(self.P, self.rescale(x01), z, alpha, self.cfg)
        self.g_opt.apply(self.P, grads_of(g_loss, self.P, 'generator/'))
        return {'g_loss': float(g_loss.detach()), 'd_loss': float(d_loss.detach())}

    def train_func(self, batches, zs, alphas):
        n = self.args.n_disc_train
        if self.cfg.model == 'gan':
            g_loss, d_loss = losses(self.P, self.rescale(batches[0]), zs[0], None, self.cfg)
            dg = grads_of(d_loss, self.P, 'discriminator/')
            gg = grads_of(g_loss, self.P, 'generator/')
            self.d_opt.apply(self.P, dg)
            self.g_opt.apply(self.P, gg)
            return {'g_loss': float(g_loss.detach()), 'd_loss': float(d_loss.detach())}
        for i in range(n):
            self.d_step(batches[i], zs[i], alphas[i])
        return self.g_step(batches[n], zs[n], alphas[n])
