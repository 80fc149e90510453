"""Pins the hand-derived NumPy oracle (incl. the gradient penalty's second-order term) against
the independent torch-autograd statement, in float64, for the three models."""
import numpy as np
import pytest
import torch

from oracle import gan_ref as G
from oracle import tf_ops as T
from oracle import torch_ref as TR


@pytest.mark.parametrize('model', ['iwgan', 'wgan', 'gan'])
@pytest.mark.parametrize('shape', [(32, 32, 3), (64, 64, 3), (32, 32, 1)])
def test_losses_and_all_gradients(model, shape):
    if shape == (64, 64, 3) and model != 'iwgan':
        pytest.skip('one literal-64 case is enough')
    B, L = 3, 8
    cfg = G.make_cfg(model, shape, L, B)
    P = G.init_params(cfg, 0, np.float64)
    rng = np.random.default_rng(1)
    n = shape[0] * shape[1] * shape[2]
    x, z, a = rng.uniform(-1, 1, (B, n)), rng.standard_normal((B, L)), rng.uniform(0, 1, (B, 1))
    dl, dg, _ = G.d_loss_and_grads(P, x, z, a, cfg)
    gl, gg, _ = G.g_loss_and_grads(P, z, cfg)
    Pt = TR.to_torch(P, torch.float64)
    tgl, tdl = TR.losses(Pt, torch.tensor(x), torch.tensor(z), torch.tensor(a), cfg)
    assert np.allclose(dl, float(tdl.detach()), rtol=1e-10) and np.allclose(gl, float(tgl.detach()), rtol=1e-10)
    tdg = TR.grads_of(tdl, Pt, 'discriminator/')
    tgg = TR.grads_of(tgl, Pt, 'generator/')
    assert set(dg) == set(tdg) and set(gg) == set(tgg)
    for got, ref in ((dg, tdg), (gg, tgg)):
        for k in got:
            r = ref[k].numpy()
            assert np.abs(got[k] - r).max() <= 1e-9 * max(1.0, np.abs(r).max()), k


def test_torch_primitives_match_numpy_primitives():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 9, 7, 3))
    K = rng.standard_normal((5, 5, 3, 4))
    assert np.allclose(TR.conv2d_same(torch.tensor(x), torch.tensor(K), 2).numpy(), T.conv2d(x, K, 2))
    Kt = rng.standard_normal((5, 5, 6, 3))
    y = TR.conv2d_transpose_same(torch.tensor(x), torch.tensor(Kt), 2).numpy()
    assert np.allclose(y, T.conv2d_transpose(x, Kt, (2, 18, 14, 6), 2))
    K4 = rng.standard_normal((4, 4, 3, 5))           # pix2pix k4 s2
    assert np.allclose(TR.conv2d_same(torch.tensor(x[:, :8, :6]), torch.tensor(K4), 2).numpy(), T.conv2d(x[:, :8, :6], K4, 2))


def test_trainers_agree_over_two_iterations():
    from types import SimpleNamespace
    B, L = 2, 8
    cfg = G.make_cfg('iwgan', (32, 32, 3), L, B)
    P = G.init_params(cfg, 3, np.float64)
    args = SimpleNamespace(optimizer='adam', lr=1e-3, beta1=0.5, beta2=0.9, decay=0.9, momentum=0.01, centered=False,
                           n_disc_train=2)
    a = G.GanTrainer({k: v.copy() for k, v in P.items()}, cfg, args)
    b = TR.TorchGanTrainer(TR.to_torch(P, torch.float64), cfg, args)
    rng = np.random.default_rng(4)
    for _ in range(2):
        xs = [rng.uniform(0, 1, (B, 32, 32, 3)) for _ in range(3)]
        zs = [rng.standard_normal((B, L)) for _ in range(3)]
        als = [rng.uniform(0, 1, (B, 1)) for _ in range(3)]
        ra = a.train_func(xs, zs, als)
        rb = b.train_func([torch.tensor(v) for v in xs], [torch.tensor(v) for v in zs], [torch.tensor(v) for v in als])
        assert np.allclose(ra['g_loss'], rb['g_loss'], rtol=1e-6, atol=1e-8)
        assert np.allclose(ra['d_loss'], rb['d_loss'], rtol=1e-6, atol=1e-8)
