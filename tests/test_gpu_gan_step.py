"""GPU parity of whole D / G steps (models/gan.py semantics) against the NumPy oracle on
identical injected inputs (weights, batch, z, alpha): losses, every gradient, post-step weights.

Tolerance: the north-star's 1e-3 (relative to each tensor's max magnitude) on the f32 path;
the bf16 path is checked for direction only (cosine similarity), it is the throughput path.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import gan_ref as G

pytestmark = pytest.mark.gpu


def make_args(model, B, L, shape, optimizer='adam'):
    return SimpleNamespace(model=model, batch_size=B, latent_size=L, image_shape=shape, n_gpus=1,
                           optimizer=optimizer, lr=1e-3, beta1=0.5, beta2=0.9, decay=0.9, momentum=0.01,
                           centered=False, n_disc_train=2, display_d_loss=True)


class ListSource:
    def __init__(self, batches, device):
        self.batches, self.device, self.i = batches, device, 0

    def next_batch(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return torch.tensor(b, dtype=torch.float32, device=self.device)


def relerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def relerr_where_significant(new, ref, grad_ref):
    """Adam's first step is lr*sign(g) for EVERY element, so elements whose true gradient is
    (numerically) zero move by +-lr on rounding noise alone, in the oracle as well; compare the
    updated weights only where the gradient is significant."""
    m = np.abs(grad_ref) > 1e-3 * np.abs(grad_ref).max()
    if not m.any():
        return 0.0
    return float(np.abs(np.asarray(new, np.float64) - ref)[m].max() / (np.abs(ref).max() + 1e-30))


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


def build(model, dtype, B=4, L=8, shape=(32, 32, 3), optimizer='adam', seed=0):
    gan = pkg('models.gan')
    rt = pkg('runtime')
    dev = torch.device('cuda:0')
    args = make_args(model, B, L, shape, optimizer)
    cfg = G.make_cfg(model, shape, L, B)
    P = G.init_params(cfg, seed, np.float64)
    rng = np.random.default_rng(seed + 1)
    n_steps = args.n_disc_train + 1
    batches = [rng.uniform(0, 1, (B,) + shape).astype(np.float32) for _ in range(n_steps)]
    zs = [rng.standard_normal((B, L)).astype(np.float32) for _ in range(n_steps)]
    alphas = [rng.uniform(0, 1, (B, 1)).astype(np.float32) for _ in range(n_steps)]
    sess = rt.Session(device=dev, dtype=dtype, seed=seed, rank=0, world_size=1)
    rep = gan.GanReplica(ListSource(batches, dev), args, sess)
    rep.load_variables({k: v.astype(np.float32) for k, v in P.items()})
    return args, cfg, P, batches, zs, alphas, sess, rep


@pytest.mark.parametrize('model', ['iwgan', 'wgan'])
def test_d_and_g_step_f32(model):
    args, cfg, P, batches, zs, alphas, sess, rep = build(model, 0)
    tr = G.GanTrainer({k: v.copy() for k, v in P.items()}, cfg, args)
    # ---- D step
    sess.inject = {'z': [zs[0]], 'alpha': [alphas[0]]}
    rep.d_step(rep.x_source.next_batch())
    x = tr.rescale(batches[0].astype(np.float64))
    loss, grads, aux = G.d_loss_and_grads(P, x, zs[0].astype(np.float64), alphas[0].astype(np.float64), cfg)
    got = rep.gradients()
    for k, g in grads.items():
        if k.endswith('/bias') and cfg.d_bn and ('/c2/' in k or '/c3/' in k):
            continue                       # bias under batch norm: exactly-zero gradient, pure rounding noise
        assert relerr(got[k], g) < 1e-3, k
    s = rep.scal.cpu().numpy()
    assert abs(s[rep.S_DREAL] - aux['d_real'].mean()) < 1e-4
    assert abs(s[rep.S_DFAKE] - aux['d_fake'].mean()) < 1e-4
    if model == 'iwgan':
        assert abs(s[rep.S_GP] - aux['gp']) < 1e-3 * max(1.0, aux['gp'])
    tr.d_step(batches[0].astype(np.float64), zs[0].astype(np.float64), alphas[0].astype(np.float64))
    new = rep.variables()
    for k in grads:
        if k.endswith('/bias') and cfg.d_bn and ('/c2/' in k or '/c3/' in k):
            continue
        assert relerr_where_significant(new[k], tr.P[k], grads[k]) < 1e-3, k
    # ---- G step on the updated D
    sess.inject = {'z': [zs[1]], 'alpha': [alphas[1]]}
    rep.g_step(rep.x_source.next_batch())
    gl, ggrads, _ = G.g_loss_and_grads(tr.P, zs[1].astype(np.float64), cfg)
    got = rep.gradients()
    for k, g in ggrads.items():
        if k.endswith('/bias') and 'dc4' not in k:
            continue                       # biases feeding batch norm (zero gradient up to rounding)
        assert relerr(got[k], g) < 1e-3, k
    ref = tr.g_step(batches[1].astype(np.float64), zs[1].astype(np.float64), alphas[1].astype(np.float64))
    out = rep.losses()
    assert abs(out['g_loss'] - ref['g_loss']) < 1e-3 * max(1, abs(ref['g_loss']))
    assert abs(out['d_loss'] - ref['d_loss']) < 1e-3 * max(1, abs(ref['d_loss']))
    new = rep.variables()
    for k in ggrads:
        if k.endswith('/bias') and 'dc4' not in k:
            continue                       # zero-gradient variables: Adam normalises pure rounding noise to +-lr
        assert relerr_where_significant(new[k], tr.P[k], ggrads[k]) < 1e-3, k


@pytest.mark.parametrize('model', ['iwgan', 'wgan'])
def test_train_func_sequence_f32(model):
    """Three full train_func calls ((n_disc_train + 1) batches each) track the oracle."""
    args, cfg, P, batches, zs, alphas, sess, rep = build(model, 0, optimizer='rmsprop')
    tr = G.GanTrainer({k: v.copy() for k, v in P.items()}, cfg, args)
    n = args.n_disc_train + 1
    for it in range(3):
        sess.inject = {'z': [z for z in zs], 'alpha': [a for a in alphas]}
        if model != 'iwgan':
            sess.inject.pop('alpha')
        out = rep.train_func()
        ref = tr.train_func([b.astype(np.float64) for b in batches], [z.astype(np.float64) for z in zs],
                            [a.astype(np.float64) for a in alphas])
        assert abs(out['g_loss'] - ref['g_loss']) < 2e-3 * max(1, abs(ref['g_loss'])), (it, out, ref)
        assert abs(out['d_loss'] - ref['d_loss']) < 2e-3 * max(1, abs(ref['d_loss'])), (it, out, ref)
    assert set(out) == {'g_loss', 'd_loss'}
    assert sess.global_step == 3 * n


def test_d_step_bf16_direction():
    args, cfg, P, batches, zs, alphas, sess, rep = build('iwgan', 1, B=8, L=16)
    sess.inject = {'z': [zs[0]], 'alpha': [alphas[0]]}
    rep.d_step(rep.x_source.next_batch())
    tr = G.GanTrainer(P, cfg, args)
    x = tr.rescale(batches[0].astype(np.float64))
    _, grads, _ = G.d_loss_and_grads(P, x, zs[0].astype(np.float64), alphas[0].astype(np.float64), cfg)
    got = rep.gradients()
    for k in ['discriminator/vars/c1/weights', 'discriminator/vars/c2/weights', 'discriminator/vars/c3/weights',
              'discriminator/vars/fc2/weights']:
        assert cosine(got[k], grads[k]) > 0.98, k


def test_graph_replay_matches_eager():
    """hipGraph replay of the step bodies (device-resident RNG draw counter and Adam step count)
    produces bit-identical variables to eager execution over several iterations."""
    import copy
    results = []
    for use_graphs in (False, True):
        gan, rt, data, K = pkg('models.gan'), pkg('runtime'), pkg('data'), pkg('kernels')
        dev = torch.device('cuda:0')
        args = make_args('iwgan', 8, 16, (32, 32, 3), 'adam')
        args.use_graphs = use_graphs
        sess = rt.Session(device=dev, dtype=0, seed=3, rank=0, world_size=1)
        src = data.SyntheticSource(64, (32, 32, 3), 8, dev, seed=5)
        rep = gan.GanReplica(src, args, sess)
        for _ in range(4):                      # eager warm-up, capture, then two replays
            out = rep.train_func()
        results.append((rep.variables(), out, rep.d_opt.t, rep.g_opt.t))
        assert bool(rep._graphs) == use_graphs
    (va, oa, ta, tga), (vb, ob, tb, tgb) = results
    assert (ta, tga) == (tb, tgb) == (8, 4)
    assert oa == ob
    for k in va:
        assert np.array_equal(va[k], vb[k]), k


BN_FED_BIASES = {'generator/vars/%s/bias' % n for n in ('fc1', 'dc1', 'dc2', 'dc3')} | \
    {'discriminator/vars/%s/bias' % n for n in ('c2', 'c3')}


def test_vanilla_gan_step_f32():
    """--model gan: one batch, D and G gradients from the same forward (models/gan.py:110-131,193-194)."""
    args, cfg, P, batches, zs, alphas, sess, rep = build('gan', 0, optimizer='rmsprop')
    tr = G.GanTrainer({k: v.copy() for k, v in P.items()}, cfg, args)
    sess.inject = {'z': [zs[0]]}
    out = rep.train_func()
    x = tr.rescale(batches[0].astype(np.float64))
    dl, dg, _ = G.d_loss_and_grads(P, x, zs[0].astype(np.float64), None, cfg)
    gl, gg, _ = G.g_loss_and_grads(P, zs[0].astype(np.float64), cfg)
    got = rep.gradients()
    for grads in (dg, gg):
        for k, g in grads.items():
            if k in BN_FED_BIASES:
                continue                   # biases under batch norm: zero gradient up to rounding
            assert relerr(got[k], g) < 1e-3, k
    assert abs(out['d_loss'] - dl) < 1e-3 * max(1, abs(dl)) and abs(out['g_loss'] - gl) < 1e-3 * max(1, abs(gl))
    ref = tr.train_func([batches[0].astype(np.float64)], [zs[0].astype(np.float64)])
    new = rep.variables()
    for k in list(dg) + list(gg):
        if k in BN_FED_BIASES:
            continue
        assert relerr(new[k], tr.P[k]) < 1e-3, k
    assert sess.global_step == 2
