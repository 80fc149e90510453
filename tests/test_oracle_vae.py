"""Pins the hand-derived VAE oracle against the independent torch-autograd statement (float64)."""
import numpy as np
import torch

from oracle import torch_ref as TR
from oracle import vae_ref as V


def test_vae_losses_and_gradients_match_autograd():
    L, B = 8, 2
    P = V.init_params(L, 0, np.float64)
    rng = np.random.default_rng(1)
    x, eps = rng.uniform(0, 1, (B, 64, 64, 3)), rng.standard_normal((B, L))
    losses, c = V.forward(P, x, eps)
    g = V.backward(P, c)
    Pt = TR.to_torch(P, torch.float64)
    dl, ll = V.torch_losses(Pt, torch.tensor(x), torch.tensor(eps))
    assert np.allclose(losses['decoder_loss'], float(dl.detach())) and np.allclose(losses['latent_loss'], float(ll.detach()))
    gs = torch.autograd.grad(dl, list(Pt.values()), allow_unused=True)
    assert set(g) == set(Pt)
    for (k, v), gt in zip(Pt.items(), gs):
        gt = np.zeros(v.shape) if gt is None else gt.numpy()
        assert np.abs(g[k] - gt).max() <= 1e-8 * max(1.0, np.abs(gt).max()), k


def test_cnn_oracle_loss_is_l1_of_rescaled_input():
    """models/cnn.py:31,75-79: loss = mean |2(x-0.5) - d|, d in (-1,1) (tanh); one dense latent, no batch norm."""
    from oracle import cnn_ref as CR
    L, B = 8, 2
    P = TR.to_torch(CR.init_params(L, 0, np.float64), torch.float64)
    assert 'encoder/BatchNorm/beta' not in P and 'latent/vars/d2/weights' not in P
    x = torch.tensor(np.random.default_rng(0).uniform(0, 1, (B, 64, 64, 3)))
    loss, d = CR.forward(P, x)
    assert d.shape == x.shape and float(d.detach().abs().max()) < 1.0
    assert np.allclose(float(loss.detach()), float((2 * (x - 0.5) - d.detach()).abs().mean()))
    g = torch.autograd.grad(loss, list(P.values()))
    assert all(torch.isfinite(t).all() and t.abs().sum() > 0 for t in g)
