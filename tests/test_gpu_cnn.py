"""GPU parity of the convolutional autoencoder step (models/cnn.py semantics) against the torch-autograd oracle."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import cnn_ref as CR
from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def test_cnn_step_f32():
    """Loss, every gradient and the post-step variables of two consecutive steps: f32 HIP path vs float64 oracle, 1e-3."""
    cnn, rt, data, K = pkg('models.cnn'), pkg('runtime'), pkg('data'), pkg('kernels')
    dev = torch.device('cuda:0')
    B, L = 3, 8
    args = SimpleNamespace(model='cnn', batch_size=B, latent_size=L, image_shape=(64, 64, 3), n_gpus=1, optimizer='rmsprop',
                           lr=1e-3, decay=0.9, momentum=0.01, centered=False, beta1=0.9, beta2=0.999)
    P0 = CR.init_params(L, 0, np.float64)
    rng = np.random.default_rng(2)
    xs = [rng.uniform(0, 1, (B, 64, 64, 3)).astype(np.float32) for _ in range(2)]
    sess = rt.Session(device=dev, dtype=K.F32, seed=0, rank=0, world_size=1)
    rep = cnn.CnnReplica(data.ArraySource(np.concatenate(xs), B, dev), args, sess)
    assert set(rep.store.index) == set(CR.param_shapes(L))
    rep.load_variables({k: v.astype(np.float32) for k, v in P0.items()})
    tr = CR.CnnTrainer(TR.to_torch(P0, torch.float64), args)
    for it in range(2):
        x = torch.tensor(xs[it], dtype=torch.float64)
        loss, grads = tr.loss_and_grads(x)
        out = rep.train_func()
        got = rep.gradients()
        for k, g in grads.items():
            assert relerr(got[k], g.numpy()) < 1e-3, (it, k)
        assert abs(out['loss'] - loss) < 1e-4 * max(1.0, abs(loss)), (it, out, loss)
        tr.train_func(x)
        new = rep.variables()
        for k in grads:
            assert relerr(new[k], tr.P[k].detach().numpy()) < 1e-3, (it, k)
    assert set(out) == {'loss'}
    real, recon = rep.samples(2)
    assert real.shape == recon.shape == (2, 64, 64, 3) and np.abs(recon).max() <= 1.0


def test_cnn_bf16_runs_and_loss_decreases():
    cnn, rt, data, K = pkg('models.cnn'), pkg('runtime'), pkg('data'), pkg('kernels')
    dev = torch.device('cuda:0')
    args = SimpleNamespace(model='cnn', batch_size=16, latent_size=32, image_shape=(64, 64, 3), n_gpus=1, optimizer='adam',
                           lr=1e-3, decay=0.9, momentum=0.01, centered=False, beta1=0.9, beta2=0.999)
    sess = rt.Session(device=dev, dtype=K.BF16, seed=1, rank=0, world_size=1)
    rep = cnn.CnnReplica(data.SyntheticSource(16, (64, 64, 3), 16, dev, seed=3), args, sess)
    first = rep.train_func()['loss']
    for _ in range(30):
        last = rep.train_func()['loss']
    assert np.isfinite(last) and last < first
