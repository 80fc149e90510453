"""GPU parity of the VAE step (models/vae.py semantics) against the NumPy oracle on injected inputs."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import vae_ref as V

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def test_vae_step_f32():
    vae, rt, data, K = pkg('models.vae'), pkg('runtime'), pkg('data'), pkg('kernels')
    dev = torch.device('cuda:0')
    B, L = 3, 8
    args = SimpleNamespace(model='vae', batch_size=B, latent_size=L, image_shape=(64, 64, 3), n_gpus=1, optimizer='rmsprop',
                           lr=1e-3, decay=0.9, momentum=0.01, centered=False, beta1=0.9, beta2=0.999)
    P = V.init_params(L, 0, np.float64)
    rng = np.random.default_rng(2)
    xs = [rng.uniform(0, 1, (B, 64, 64, 3)).astype(np.float32) for _ in range(2)]
    eps = [rng.standard_normal((B, L)).astype(np.float32) for _ in range(2)]
    sess = rt.Session(device=dev, dtype=K.F32, seed=0, rank=0, world_size=1)
    rep = vae.VaeReplica(data.ArraySource(np.concatenate(xs), B, dev), args, sess)
    assert set(rep.store.index) == set(V.param_shapes(L))
    rep.load_variables({k: v.astype(np.float32) for k, v in P.items()})
    tr = V.VaeTrainer({k: v.copy() for k, v in P.items()}, args)
    for it in range(2):
        sess.inject = {'eps': [eps[it]]}
        losses, c = V.forward(tr.P, xs[it].astype(np.float64), eps[it].astype(np.float64))
        grads = V.backward(tr.P, c)
        out = rep.train_func()
        got = rep.gradients()
        for k, g in grads.items():
            if k.startswith('encoder/vars/') and k.endswith('/bias'):
                continue                     # biases feeding batch norm: zero gradient up to rounding
            assert relerr(got[k], g) < 1e-3, (it, k)
        for k in ('decoder_loss', 'latent_loss', 'total_loss'):
            assert abs(out[k] - losses[k]) < 1e-3 * max(1.0, abs(losses[k])), (it, k, out[k], losses[k])
        tr.train_func(xs[it].astype(np.float64), eps[it].astype(np.float64))
        new = rep.variables()
        for k in grads:
            if k.startswith('encoder/vars/') and k.endswith('/bias'):
                continue
            assert relerr(new[k], tr.P[k]) < 1e-3, (it, k)
    assert set(out) == {'decoder_loss', 'latent_loss', 'total_loss'}


def test_vae_bf16_runs_and_loss_decreases():
    vae, rt, data, K = pkg('models.vae'), pkg('runtime'), pkg('data'), pkg('kernels')
    dev = torch.device('cuda:0')
    args = SimpleNamespace(model='vae', batch_size=16, latent_size=32, image_shape=(64, 64, 3), n_gpus=1, optimizer='adam',
                           lr=1e-3, decay=0.9, momentum=0.01, centered=False, beta1=0.9, beta2=0.999)
    sess = rt.Session(device=dev, dtype=K.BF16, seed=1, rank=0, world_size=1)
    rep = vae.VaeReplica(data.SyntheticSource(16, (64, 64, 3), 16, dev, seed=3), args, sess)
    first = rep.train_func()['decoder_loss']
    for _ in range(30):
        last = rep.train_func()['decoder_loss']
    assert np.isfinite(last) and last < first


def test_vae_graph_replay_matches_eager():
    """The captured step bodies (hipGraph replay) reproduce the eager steps bit for bit: same seeds, same batches."""
    vae, rt, data, K = pkg('models.vae'), pkg('runtime'), pkg('data'), pkg('kernels')
    dev = torch.device('cuda:0')
    res = []
    for graphs in (False, True):
        args = SimpleNamespace(model='vae', batch_size=8, latent_size=16, image_shape=(64, 64, 3), n_gpus=1, optimizer='adam',
                               lr=1e-3, decay=0.9, momentum=0.01, centered=False, beta1=0.9, beta2=0.999, use_graphs=graphs)
        sess = rt.Session(device=dev, dtype=K.BF16, seed=5, rank=0, world_size=1)
        rep = vae.VaeReplica(data.SyntheticSource(32, (64, 64, 3), 8, dev, seed=3), args, sess)
        losses = [rep.train_func()['decoder_loss'] for _ in range(5)]
        assert bool(rep._graphs) == graphs
        res.append((losses, {k: v.copy() for k, v in rep.variables().items()}))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert np.array_equal(res[0][1][k], res[1][1][k]), k


def test_vae_full_elbo_opt_in_f32():
    """--vae_full_elbo (SURVEY App. C-7 opt-in): gradients of decoder_loss + latent_loss; the reference (and the default
    here) differentiates decoder_loss alone (models/vae.py:41).  Only the paths through the heads change."""
    vae, rt, data, K = pkg('models.vae'), pkg('runtime'), pkg('data'), pkg('kernels')
    dev = torch.device('cuda:0')
    B, L = 3, 8
    args = SimpleNamespace(model='vae', batch_size=B, latent_size=L, image_shape=(64, 64, 3), n_gpus=1, optimizer='rmsprop',
                           lr=1e-3, decay=0.9, momentum=0.01, centered=False, beta1=0.9, beta2=0.999, vae_full_elbo=True)
    P = V.init_params(L, 0, np.float64)
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 1, (B, 64, 64, 3)).astype(np.float32)
    eps = rng.standard_normal((B, L)).astype(np.float32)
    sess = rt.Session(device=dev, dtype=K.F32, seed=0, rank=0, world_size=1)
    rep = vae.VaeReplica(data.ArraySource(x, B, dev), args, sess)
    rep.load_variables({k: v.astype(np.float32) for k, v in P.items()})
    sess.inject = {'eps': [eps]}
    losses, c = V.forward(P, x.astype(np.float64), eps.astype(np.float64))
    full, recon = V.backward(P, c, full_elbo=True), V.backward(P, c)
    rep.train_func()
    got = rep.gradients()
    moved = 0
    for k, g in full.items():
        if k.startswith('encoder/vars/') and k.endswith('/bias'):
            continue
        assert relerr(got[k], g) < 1e-3, k
        moved += relerr(recon[k], g) > 1e-3
    assert moved >= 4                                   # the latent heads and the encoder see the KL term; the decoder does not
    # the independent autograd statement of the same sum
    import torch as th
    Pt = {k: th.tensor(v, dtype=th.float64, requires_grad=True) for k, v in P.items()}
    d_loss, l_loss = V.torch_losses(Pt, th.tensor(x, dtype=th.float64), th.tensor(eps, dtype=th.float64))
    keys = [k for k in full if not k.startswith('encoder/BatchNorm')]
    auto = th.autograd.grad(d_loss + l_loss, [Pt[k] for k in keys], allow_unused=True)
    for k, a in zip(keys, auto):
        if k.startswith('encoder/vars/') and k.endswith('/bias'):
            continue
        assert relerr(full[k], a.numpy()) < 1e-8, k
