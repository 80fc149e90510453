"""CPU restatement (torch autograd, float64-capable) of the reference's pix2pix cGAN
(`hem/models/pix2pix.py:81-304` with `hem/ops/layers.py:70-211`, `hem/ops/images.py:53-70`,
`hem/ops/losses.py:10-11`).

TEST INFRASTRUCTURE ONLY (see oracle/tf_ops.py).  PARITY UNPINNED against TensorFlow.  Unlike
gan_ref / vae_ref there is no second, hand-derived statement of this model: the conv / deconv /
batch-norm / lrelu primitives used here are the ones pinned in tests/test_oracle_vs_autograd.py and the
known-answer tests; the model-level statement below is autograd over those primitives.

Layout is NHWC here (the reference runs NCHW, `data_format='NCHW'`; TF filters are HWIO in both, so the
variables are identical and only the activation layout differs).

Effective semantics reproduced: skip connections always on; `--lambda` ignored
(L1 weight hard-coded 10.0, pix2pix.py:286); batch norm on EVERY decoder layer including the final
1-channel one before tanh (:196-227); decoder activation lrelu(leak=0) == relu; init N(0, 0.02) for weights
and biases (:180,201,248); losses dict keyed by the last name component (two tensors are called 'total':
the discriminator's wins; with --add_l1 the generator term is the tensor '.../add').
"""
import numpy as np
import torch

from . import torch_ref as TR

ENC = [(3, 64), (64, 128), (128, 256), (256, 512), (512, 512), (512, 512), (512, 512), (512, 512)]   # pix2pix.py:185-194
DEC = [(512, 512), (1024, 512), (1024, 512), (1024, 512), (1024, 256), (512, 128), (256, 64), (128, 1)]  # :207-227
DISC = [(4, 64), (64, 128), (128, 256), (256, 512), (512, 1)]                                          # :252-256


def _bn(scope, i):
    return '%s/BatchNorm/beta' % scope if i == 0 else '%s/BatchNorm_%d/beta' % (scope, i)


def _widths(args):
    """Layer widths with `--noise` (hem/models/pix2pix.py:183-186,204-206,223-225): one more input channel on encoder 1,
    512 more on decoder 1, one more on decoder 8."""
    noise = getattr(args, 'noise', None) or []
    enc, dec = list(ENC), list(DEC)
    if 'input' in noise:
        enc[0] = (4, 64)
    if 'latent' in noise:
        dec[0] = (1024, 512)
    if 'end' in noise:
        dec[7] = (129, 1)
    return enc, dec


def param_shapes(args):
    sh = {}
    nb = 0
    ENC, DEC = _widths(args)
    for i, (ci, co) in enumerate(ENC, start=1):
        sh['generator/enocder/vars/%d/weights' % i] = (4, 4, ci, co)
        sh['generator/enocder/vars/%d/bias' % i] = (co,)
        if args.batch_norm_gen and i > 1:
            sh[_bn('generator/enocder', nb)] = (co,)
            nb += 1
    for i, (ci, co) in enumerate(DEC, start=1):
        sh['generator/decoder/vars/%d/weights' % i] = (4, 4, co, ci)
        sh['generator/decoder/vars/%d/bias' % i] = (co,)
        sh[_bn('generator/decoder', i - 1)] = (co,)
    for i, (ci, co) in enumerate(DISC, start=1):
        sh['discriminator/vars/m%d/weights' % i] = (4, 4, ci, co)
        sh['discriminator/vars/m%d/bias' % i] = (co,)
    if args.batch_norm_disc:                       # m2..m5: the arg_scope default also reaches the logits layer (:246-256)
        for p in range(2):
            for li, (ci, co) in enumerate(DISC[1:5]):
                sh[_bn('discriminator', p * 4 + li)] = (co,)
    return sh


def init_params(args, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return {k: (np.zeros(s, dtype) if k.endswith('/beta') else (rng.standard_normal(s) * 0.02).astype(dtype))
            for k, s in param_shapes(args).items()}


def generator(P, x, args, drops=None, noise=None):
    """x: [B,256,256,3] in [-1,1] -> [B,256,256,1] (tanh).  `drops`: the uniform draws [B,h,w,512] of the dropout on
    decoder layers 1-3 (hem/models/pix2pix.py:204-208 `dropout=args.dropout`; hem/ops/layers.py:207
    `tf.nn.dropout(h, keep_prob=dropout)` = h * floor(keep_prob + u) / keep_prob, after the activation).
    `noise`: {'input': [B,256,256,1], 'latent': [B,1,1,512], 'end': [B,128,128,1]} draws of tf.random_uniform(-1, 1)
    for the points named in args.noise (:183-186,204-206,223-225), concatenated as trailing channels."""
    which = getattr(args, 'noise', None) or []
    e, h, nb = [], x, 0
    if 'input' in which:
        h = torch.cat([h, noise['input']], dim=-1)
    for i in range(1, 9):
        h = TR.conv2d_same(h, P['generator/enocder/vars/%d/weights' % i], 2) + P['generator/enocder/vars/%d/bias' % i]
        if args.batch_norm_gen and i > 1:
            h = TR.batch_norm(h, P[_bn('generator/enocder', nb)])
            nb += 1
        h = TR.lrelu(h, 0.2)
        e.append(h)
    y = e[7]
    if 'latent' in which:
        y = torch.cat([y, noise['latent']], dim=-1)
    for i in range(1, 9):
        if i > 1:
            y = torch.cat([y, e[8 - i]], dim=-1)                                   # tf.concat([y, e_k], axis=1) in NCHW
        if i == 8 and 'end' in which:
            y = torch.cat([y, noise['end']], dim=-1)
        y = TR.conv2d_transpose_same(y, P['generator/decoder/vars/%d/weights' % i]) + P['generator/decoder/vars/%d/bias' % i]
        y = TR.batch_norm(y, P[_bn('generator/decoder', i - 1)])
        y = torch.tanh(y) if i == 8 else torch.relu(y)
        if i <= 3 and getattr(args, 'dropout', 0) > 0:
            y = y * torch.floor(args.dropout + drops[i - 1]) / args.dropout
    return y


def discriminator(P, x, y, args, bn_pass=0):
    h = torch.cat([x, y], dim=-1)
    for i in range(1, 6):
        h = TR.conv2d_same(h, P['discriminator/vars/m%d/weights' % i], 2) + P['discriminator/vars/m%d/bias' % i]
        if args.batch_norm_disc and i >= 2:
            h = TR.batch_norm(h, P[_bn('discriminator', bn_pass * 4 + (i - 2))])
        if i < 5:
            h = TR.lrelu(h, 0.2)
    return h                                                                        # logits [B,8,8,1]


def xent(logits, label):
    return torch.clamp(logits, min=0) - logits * label + torch.log1p(torch.exp(-torch.abs(logits)))


def losses(P, x01, y01, args):
    """hem/models/pix2pix.py:101-120,263-304 on one batch; returns (g_total, d_total, report dict)."""
    x, y = 2 * x01 - 1, 2 * y01 - 1                                                 # hem.rescale((0,1)->(-1,1))
    g = generator(P, x, args)
    d_real_logits = discriminator(P, x, y, args, 0)
    d_fake_logits = discriminator(P, x, g, args, 1)
    g01, yy = (g + 1) / 2, (y + 1) / 2
    g_fake = xent(d_fake_logits, 1.0).mean()
    l1 = (yy - g01).abs().mean()
    g_total = g_fake + 10.0 * l1 if args.add_l1 else g_fake
    d_real = xent(d_real_logits, 1.0).mean()
    d_fake = xent(d_fake_logits, 0.0).mean()
    d_total = d_real + d_fake
    rmse = torch.sqrt(((g01 - yy) ** 2).mean())
    rep = {'l1': l1, ('add' if args.add_l1 else 'g_fake'): g_total, 'd_real': d_real, 'd_fake': d_fake, 'total': d_total, 'rmse': rmse}
    return g_total, d_total, {k: float(v.detach()) for k, v in rep.items()}


class Trainer:
    """pix2pix.train (:151-156): n_disc_train D steps, one G step, then the losses of a THIRD batch."""

    def __init__(self, P, args):
        self.P, self.args = P, args
        self.g_opt, self.d_opt = TR.make_optimizer(args), TR.make_optimizer(args)

    def d_step(self, x01, y01):
        _, d_total, _ = losses(self.P, x01, y01, self.args)
        self.d_opt.apply(self.P, TR.grads_of(d_total, self.P, 'discriminator/'))

    def g_step(self, x01, y01):
        g_total, _, _ = losses(self.P, x01, y01, self.args)
        self.g_opt.apply(self.P, TR.grads_of(g_total, self.P, 'generator/'))

    def train(self, batches):
        n = self.args.n_disc_train
        for i in range(n):
            self.d_step(*batches[i])
        self.g_step(*batches[n])
        with torch.no_grad():
            return losses(self.P, *batches[n + 1], self.args)[2]
