"""Pins oracle/towers_ref.py (the reference's multi-tower step: util.py:54-77,118-147; models/gan.py:55-81):
* one tower == the single-replica oracle trainer;
* the mean of the per-tower gradients == torch autograd of the MEAN of the per-tower losses (independent statement:
  averaging is linear, so d/dP [1/n sum_i loss_i] must come out variable by variable), with per-tower batch norm and a
  per-tower whole-batch penalty norm (NOT the loss of the concatenated batch);
* two towers on different shards differ from one tower on either shard (the rehearsal inputs are able to tell)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import gan_ref as G
from oracle import torch_ref as TR
from oracle import towers_ref as TW


def _args(opt='adam'):
    return SimpleNamespace(optimizer=opt, lr=1e-3, beta1=0.5, beta2=0.9, decay=0.9, momentum=0.01, centered=False, n_disc_train=2)


def _inputs(rng, B, L, shape, n):
    return ([rng.uniform(0, 1, (B,) + shape) for _ in range(n)], [rng.standard_normal((B, L)) for _ in range(n)],
            [rng.uniform(0, 1, (B, 1)) for _ in range(n)])


@pytest.mark.parametrize('model', ['iwgan', 'wgan'])
def test_one_tower_is_the_single_replica_trainer(model):
    B, L, shape = 2, 8, (32, 32, 3)
    cfg = G.make_cfg(model, shape, L, B)
    P = G.init_params(cfg, 3, np.float64)
    a = G.GanTrainer({k: v.copy() for k, v in P.items()}, cfg, _args())
    b = TW.GanTowers({k: v.copy() for k, v in P.items()}, cfg, _args())
    xs, zs, als = _inputs(np.random.default_rng(5), B, L, shape, 3)
    ref = a.train_func(xs, zs, als)
    for i in range(2):
        b.d_step([xs[i]], [zs[i]], [als[i]])
    out = b.g_step([xs[2]], [zs[2]], [als[2]])
    assert out == ref
    for k in P:
        assert np.array_equal(a.P[k], b.P[k]), k


@pytest.mark.parametrize('model', ['iwgan', 'wgan'])
def test_tower_mean_is_the_gradient_of_the_mean_loss(model):
    B, L, shape, n = 3, 8, (32, 32, 3), 2
    cfg = G.make_cfg(model, shape, L, B)
    P = G.init_params(cfg, 1, np.float64)
    xs, zs, als = _inputs(np.random.default_rng(6), B, L, shape, n)
    tw = TW.GanTowers({k: v.copy() for k, v in P.items()}, cfg, _args('sgd'))
    tw.d_step(xs, zs, als)
    Pt = TR.to_torch(P, torch.float64)
    tower = [TR.losses(Pt, torch.tensor(TW.GanTowers.rescale(x)), torch.tensor(z), torch.tensor(a), cfg) for x, z, a in zip(xs, zs, als)]
    d_mean = sum(t[1] for t in tower) / n
    ref = TR.grads_of(d_mean, Pt, 'discriminator/')
    for k, r in ref.items():
        r = r.numpy()
        assert np.abs(tw.last_d_grads[k] - r).max() <= 1e-9 * max(1.0, np.abs(r).max()), k
    # and it is NOT the gradient of one replica on the concatenated batch (per-tower penalty norm / batch statistics)
    _, d_cat = TR.losses(Pt, torch.tensor(TW.GanTowers.rescale(np.concatenate(xs))), torch.tensor(np.concatenate(zs)),
                         torch.tensor(np.concatenate(als)), G.make_cfg(model, shape, L, n * B))
    cat = TR.grads_of(d_cat, Pt, 'discriminator/')
    k = 'discriminator/vars/c3/weights'
    assert np.abs(tw.last_d_grads[k] - cat[k].numpy()).max() > 1e-3 * np.abs(cat[k].numpy()).max()


def test_towers_on_different_shards_differ_from_either_shard_alone():
    B, L, shape = 2, 8, (32, 32, 3)
    cfg = G.make_cfg('iwgan', shape, L, B)
    P = G.init_params(cfg, 2, np.float64)
    xs, zs, als = _inputs(np.random.default_rng(7), B, L, shape, 2)
    both = TW.GanTowers({k: v.copy() for k, v in P.items()}, cfg, _args())
    both.d_step(xs, zs, als)
    for i in range(2):
        one = TW.GanTowers({k: v.copy() for k, v in P.items()}, cfg, _args())
        one.d_step([xs[i]], [zs[i]], [als[i]])
        k = 'discriminator/vars/c1/weights'
        assert np.abs(both.last_d_grads[k] - one.last_d_grads[k]).max() > 1e-2 * np.abs(both.last_d_grads[k]).max()


def test_vae_tower_mean():
    from oracle import vae_ref as V
    L, B = 8, 2
    P = V.init_params(L, 0, np.float64)
    rng = np.random.default_rng(8)
    xs = [rng.uniform(0, 1, (B, 64, 64, 3)) for _ in range(2)]
    es = [rng.standard_normal((B, L)) for _ in range(2)]
    tw = TW.VaeTowers({k: v.copy() for k, v in P.items()}, _args())
    rep = tw.step(xs, es)
    singles = []
    for x, e in zip(xs, es):
        losses, c = V.forward(P, x, e)
        singles.append((losses, V.backward(P, c)))
    assert rep == {k: float(v) for k, v in singles[-1][0].items()}            # the last tower's losses
    for k, g in tw.last_grads.items():
        assert np.allclose(g, 0.5 * (singles[0][1][k] + singles[1][1][k]), rtol=1e-12, atol=0), k
