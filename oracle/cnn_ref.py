"""CPU restatement of the reference's convolutional autoencoder (`models/cnn.py:20-135`) as torch autograd.

TEST INFRASTRUCTURE ONLY (see oracle/tf_ops.py).  PARITY UNPINNED against TensorFlow: TensorFlow is not
installed here and the reference holds no fixture for this model.

Semantics followed: inputs rescaled to [-1, 1] (`:31`); encoder conv k5 s2 3->64->128->256->256 then 1x1
256->96->32, lrelu 0.2, NO batch norm (`:104-119`, unlike the VAE's encoder); latent = one dense 512 -> L
(`:82-93`); decoder = dense L -> 512 relu, reshape [-1,4,4,32], 1x1 32->96->256 relu, deconv k5 s2
256->256->128->64 relu, ->3 tanh (`:122-134`); loss = mean |x - d| (`:75-79`); one optimizer over every
variable, one batch per call (`util.py:22-28`).  Hard-wired to 64x64 inputs through `32*4*4`.
"""
import numpy as np
import torch

from . import tf_ops as T
from . import torch_ref as TR

ENC = [('c1', 3, 64, 5, 2), ('c2', 64, 128, 5, 2), ('c3', 128, 256, 5, 2), ('c4', 256, 256, 5, 2),
       ('c5', 256, 96, 1, 1), ('c6', 96, 32, 1, 1)]
DEC_CONV = [('c1', 32, 96, 1, 1), ('c2', 96, 256, 1, 1)]
DEC_DECONV = [('dc1', 256, 256), ('dc2', 256, 128), ('dc3', 128, 64), ('dc4', 64, 3)]


def param_shapes(L, cin=3):
    sh = {}
    for i, (n, ci, co, k, s) in enumerate(ENC):
        ci = cin if i == 0 else ci
        sh['encoder/vars/%s/weights' % n] = (k, k, ci, co)
        sh['encoder/vars/%s/bias' % n] = (co,)
    sh['latent/vars/d1/weights'] = (512, L)
    sh['latent/vars/d1/bias'] = (L,)
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
    return {k: T.xavier_uniform(s, rng, dtype) for k, s in param_shapes(L).items()}


def forward(P, x01):
    """Returns (loss, reconstruction d in [-1,1])."""
    x = 2 * (x01 - 0.5)
    h = x
    for n, _, _, k, s in ENC:
        h = TR.lrelu(TR.conv2d_same(h, P['encoder/vars/%s/weights' % n], s) + P['encoder/vars/%s/bias' % n])
    z = h.reshape(h.shape[0], -1) @ P['latent/vars/d1/weights'] + P['latent/vars/d1/bias']
    h = torch.relu(z @ P['decoder/vars/d1/weights'] + P['decoder/vars/d1/bias']).reshape(-1, 4, 4, 32)
    for n, _, _, k, s in DEC_CONV:
        h = torch.relu(TR.conv2d_same(h, P['decoder/vars/%s/weights' % n], s) + P['decoder/vars/%s/bias' % n])
    for n, _, _ in DEC_DECONV:
        u = TR.conv2d_transpose_same(h, P['decoder/vars/%s/weights' % n]) + P['decoder/vars/%s/bias' % n]
        h = torch.tanh(u) if n == 'dc4' else torch.relu(u)
    return torch.mean(torch.abs(x - h)), h


class CnnTrainer:
    """default_training (util.py:22-28): one run of [train_op, losses] per call."""

    def __init__(self, P, args):
        self.P, self.opt = P, TR.make_optimizer(args)

    def loss_and_grads(self, x01):
        loss, _ = forward(self.P, x01)
        gs = torch.autograd.grad(loss, list(self.P.values()))
        return float(loss.detach()), dict(zip(self.P, gs))

    def train_func(self, x01):
        loss, g = self.loss_and_grads(x01)
        self.opt.apply(self.P, g)
        return {'loss': loss}
