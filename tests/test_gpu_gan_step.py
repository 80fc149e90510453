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


def _free_run(model, optimizer, lr, beta1, beta2, shape, iters, B=16, L=40):
    """HIP f32 replica, float64 oracle and float32 oracle stepped side by side on the same batches / z / alpha.
    Returns [(iteration, loss name, |hip - f64|, |torch f32 - f64|)] relative to max(1, |loss|), and the last loss dict."""
    from oracle import torch_ref as TR
    gan, rt = pkg('models.gan'), pkg('runtime')
    dev = torch.device('cuda:0')
    args = SimpleNamespace(model=model, batch_size=B, latent_size=L, image_shape=shape, n_gpus=1, optimizer=optimizer,
                           lr=lr, beta1=beta1, beta2=beta2, decay=0.9, momentum=0.01, centered=False, n_disc_train=5,
                           display_d_loss=True)
    cfg = G.make_cfg(model, shape, L, B)
    P = G.init_params(cfg, 3, np.float64)
    rng = np.random.default_rng(11)
    n = args.n_disc_train + 1
    batches = [rng.uniform(0, 1, (B,) + shape).astype(np.float32) for _ in range(iters * n)]
    zs = [rng.standard_normal((B, L)).astype(np.float32) for _ in range(iters * n)]
    als = [rng.uniform(0, 1, (B, 1)).astype(np.float32) for _ in range(iters * n)]
    sess = rt.Session(device=dev, dtype=0, seed=0, rank=0, world_size=1)
    rep = gan.GanReplica(ListSource(batches, dev), args, sess)
    rep.load_variables({k: v.astype(np.float32) for k, v in P.items()})
    tr64 = TR.TorchGanTrainer(TR.to_torch(P, torch.float64), cfg, args)
    tr32 = TR.TorchGanTrainer(TR.to_torch(P, torch.float32), cfg, args)
    hist = []
    for it in range(iters):
        sl = slice(it * n, (it + 1) * n)
        sess.inject = {'z': list(zs[sl])}
        if model == 'iwgan':
            sess.inject['alpha'] = list(als[sl])
        out = rep.train_func()
        ref = tr64.train_func([torch.tensor(b, dtype=torch.float64) for b in batches[sl]],
                              [torch.tensor(z, dtype=torch.float64) for z in zs[sl]],
                              [torch.tensor(a, dtype=torch.float64) for a in als[sl]])
        r32 = tr32.train_func([torch.tensor(b) for b in batches[sl]], [torch.tensor(z) for z in zs[sl]],
                              [torch.tensor(a) for a in als[sl]])
        for k in ('g_loss', 'd_loss'):
            scale = max(1.0, abs(ref[k]))
            assert np.isfinite(out[k]), (it, k, out[k])
            hist.append((it, k, abs(out[k] - ref[k]) / scale, abs(r32[k] - ref[k]) / scale))
    assert sess.global_step == iters * n
    return hist, out


def _check_free_run(hist, exact_iters):
    """|hip - f64| < 1e-3 outright for the first `exact_iters` iterations; afterwards within max(1e-3, 5 x the largest
    |torch f32 - f64| seen so far): float32 rounding is amplified from step to step in ANY float32 implementation, the
    oracle's own float32 run is the yardstick for how far a float32 trajectory may be from the float64 one."""
    worst_sens = 0.0
    for it, k, err, sens in hist:
        worst_sens = max(worst_sens, sens)
        assert err < (1e-3 if it < exact_iters else max(1e-3, 5.0 * worst_sens)), (it, k, err, sens, worst_sens)


def test_iwgan_20_iterations_track_the_oracle():
    """The headline schedule (iwgan, adam 1e-4 / 0.5 / 0.9, n_disc_train 5) free-running for 20 train_func calls = 120
    optimizer steps on fresh batches with injected z and alpha (widths reduced to L=40, batch 16, so the oracle finishes
    in seconds).  Measured: |HIP f32 - oracle f64| 1e-8 .. 1e-3 over the first 5 iterations (the summation order of the
    reductions decides the last digit: 4e-4 .. 1.2e-3 at iteration 4 across builds), then both float32 runs
    (HIP and the oracle's own) drift from float64 together, 1e-3 .. 2e-2 by iteration 18."""
    hist, out = _free_run('iwgan', 'adam', 1e-4, 0.5, 0.9, (32, 32, 3), 20)
    print('iteration, loss, |hip-f64|, |torch f32-f64| (relative): ' + '; '.join('%d %s %.1e %.1e' % h for h in hist[::6]))
    assert set(out) == {'g_loss', 'd_loss'}
    # Audit trail: exact_iters was 5 until the red run gpurun_out/r2_t7.log (iteration 4, |hip - f64| 1.2e-3 after round 2's
    # epilogue column partials changed the summation order of the bias gradients); yardstick: the ORACLE's own float32 run is
    # 1.9e-3 off its float64 run at that iteration, so 1e-3 is not a property of any float32 evaluation from there on.
    _check_free_run(hist, 3)


def test_wgan_mnist_like_free_run():
    """SURVEY section 8d config 1 (`--model wgan --dataset mnist` padded to 32x32x1, rmsprop defaults).  The reference's
    wgan never clips (App. C-3): the critic runs away (|loss| past 50 within a few iterations), so float32 trajectories
    leave the float64 one early -- the oracle's own float32 run by 4e-3 at the third iteration."""
    hist, out = _free_run('wgan', 'rmsprop', 1e-3, 0.9, 0.999, (32, 32, 1), 8, B=64)      # the config's own --batch_size 64
    print('iteration, loss, |hip-f64|, |torch f32-f64| (relative): ' + '; '.join('%d %s %.1e %.1e' % h for h in hist))
    assert set(out) == {'g_loss', 'd_loss'}
    _check_free_run(hist, 2)


def test_bf16_free_run_tracks_f32():
    """The timed dtype as a training run: the headline schedule (iwgan, adam 1e-4 / 0.5 / 0.9, n_disc_train 5) free-running
    for 10 train_func calls = 60 optimizer steps in bf16 and in f32 on the same batches, z and alpha (both on the HIP path).
    The reported losses of the bf16 run stay within BF16_CURVE_TOL of the f32 run's (relative to max(1, |loss|))."""
    BF16_CURVE_TOL = 2e-2          # measured: <= 5.5e-3 over the 10 iterations
    gan, rt = pkg('models.gan'), pkg('runtime')
    dev = torch.device('cuda:0')
    B, L, shape, iters = 32, 40, (32, 32, 3), 10
    rng = np.random.default_rng(21)
    n = 6
    batches = [rng.uniform(0, 1, (B,) + shape).astype(np.float32) for _ in range(iters * n)]
    zs = [rng.standard_normal((B, L)).astype(np.float32) for _ in range(iters * n)]
    als = [rng.uniform(0, 1, (B, 1)).astype(np.float32) for _ in range(iters * n)]
    curves = {}
    for dtype in (0, 1):
        args = SimpleNamespace(model='iwgan', batch_size=B, latent_size=L, image_shape=shape, n_gpus=1, optimizer='adam',
                               lr=1e-4, beta1=0.5, beta2=0.9, decay=0.9, momentum=0.01, centered=False, n_disc_train=5,
                               display_d_loss=True)
        sess = rt.Session(device=dev, dtype=dtype, seed=0, rank=0, world_size=1)
        rep = gan.GanReplica(ListSource(batches, dev), args, sess)
        rep.load_variables({k: v.astype(np.float32) for k, v in G.init_params(G.make_cfg('iwgan', shape, L, B), 3, np.float64).items()})
        out = []
        for it in range(iters):
            sl = slice(it * n, (it + 1) * n)
            sess.inject = {'z': list(zs[sl]), 'alpha': list(als[sl])}
            out.append(rep.train_func())
        curves[dtype] = out
    dev_ = [(it, k, abs(curves[1][it][k] - curves[0][it][k]) / max(1.0, abs(curves[0][it][k]))) for it in range(iters)
            for k in ('g_loss', 'd_loss')]
    print('bf16 vs f32 loss curves, |diff| / max(1, |loss|): ' + '; '.join('%d %s %.1e' % d for d in dev_))
    for it, k, d in dev_:
        assert d < BF16_CURVE_TOL, (it, k, d, curves[1][it], curves[0][it])
