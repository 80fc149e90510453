"""Golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle):
CPU leg checks the oracle still reproduces them; GPU leg checks the HIP path through the C ABI."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import pkg
from oracle import gan_ref as G
from oracle import tf_ops as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _args():
    return SimpleNamespace(optimizer='rmsprop', lr=1e-3, decay=0.9, momentum=0.01, centered=False, beta1=0.5, beta2=0.9,
                           n_disc_train=1, display_d_loss=True)


def test_conv_golden_cpu():
    z = np.load(os.path.join(GOLD, 'conv_k5s2.npz'))
    x, K, dy = (z[k].astype(np.float64) for k in ('x', 'K', 'dy'))
    assert np.allclose(T.conv2d(x, K, 2), z['y'])
    assert np.allclose(T.conv2d_backprop_input(x.shape, K, dy, 2), z['dx'])
    assert np.allclose(T.conv2d_backprop_filter(x, K.shape, dy, 2), z['dK'])
    assert np.allclose(T.conv2d_transpose(dy, K, (2, 8, 8, 8), 2), z['yt'])


@pytest.mark.parametrize('model', ['iwgan', 'wgan'])
def test_gan_step_golden_cpu(model):
    z = np.load(os.path.join(GOLD, '%s_step_L8_B4.npz' % model))
    cfg = G.make_cfg(model, (32, 32, 3), 8, 4)
    P = {k[6:]: z[k].astype(np.float64) for k in z.files if k.startswith('param/')}
    assert set(P) == set(G.param_shapes(cfg))
    tr = G.GanTrainer(P, cfg, _args())
    out = tr.train_func([z['x0'].astype(np.float64), z['x1'].astype(np.float64)],
                        [z['z0'].astype(np.float64), z['z1'].astype(np.float64)],
                        [z['alpha0'].astype(np.float64), z['alpha1'].astype(np.float64)])
    assert np.allclose(out['g_loss'], z['g_loss']) and np.allclose(out['d_loss'], z['d_loss'])
    for k in P:
        assert np.allclose(tr.P[k], z['after/' + k], rtol=1e-5, atol=1e-7), k


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['iwgan', 'wgan'])
def test_gan_step_golden_gpu(model):
    """HIP path (f32) vs the committed vectors: losses within 1e-3, first D-step gradients within 1e-3
    of each tensor's max magnitude (north-star tolerance)."""
    import torch
    z = np.load(os.path.join(GOLD, '%s_step_L8_B4.npz' % model))
    gan, rt, data, K = pkg('models.gan'), pkg('runtime'), pkg('data'), pkg('kernels')
    args = _args()
    args.model, args.batch_size, args.latent_size, args.image_shape, args.n_gpus = model, 4, 8, (32, 32, 3), 1
    sess = rt.Session(device=torch.device('cuda:0'), dtype=K.F32, seed=0, rank=0, world_size=1)
    src = data.ArraySource(np.concatenate([z['x0'], z['x1']]), 4, sess.device)
    rep = gan.GanReplica(src, args, sess)
    rep.load_variables({k[6:]: z[k] for k in z.files if k.startswith('param/')})
    sess.inject = {'z': [z['z0'], z['z1']], 'alpha': [z['alpha0'], z['alpha1']]}
    if model != 'iwgan':
        sess.inject.pop('alpha')
    rep.d_step(src.next_batch())
    got = rep.gradients()
    for k in z.files:
        if not k.startswith('dgrad/'):
            continue
        name, ref = k[6:], z[k].astype(np.float64)
        if np.abs(ref).max() < 1e-12:
            continue                                      # exactly-zero gradients (biases under BN / GP)
        assert np.abs(got[name] - ref).max() <= 1e-3 * np.abs(ref).max(), name
    rep.g_step(src.next_batch())
    out = rep.losses()
    assert abs(out['g_loss'] - float(z['g_loss'])) <= 1e-3 * max(1.0, abs(float(z['g_loss'])))
    assert abs(out['d_loss'] - float(z['d_loss'])) <= 1e-3 * max(1.0, abs(float(z['d_loss'])))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [0, 1])
def test_conv_golden_gpu(dtype):
    import torch
    K = pkg('kernels')
    z = np.load(os.path.join(GOLD, 'conv_k5s2.npz'))
    dev = torch.device('cuda:0')
    big, small = K.Act(2, 8, 8, 8, dtype, dev), K.Act(2, 4, 4, 16, dtype, dev)
    conv = K.Conv(big, small, 5, 5, 2, 1, 1)
    conv.pack(torch.tensor(z['K'], device=dev))
    tol = 2e-5 if dtype == 0 else 3e-2
    big.set(z['x'])
    conv.fwd(big.ptr(), small.ptr(), 2)
    assert np.abs(small.get() - z['y']).max() <= tol * np.abs(z['y']).max()
    small.set(z['dy'])
    out = big.like()
    conv.bwd_data(small.ptr(), out.ptr(), 2)
    assert np.abs(out.get() - z['dx']).max() <= tol * np.abs(z['dx']).max()     # == conv2d_transpose (yt)
    assert np.allclose(z['dx'], z['yt'])
    dw = torch.zeros(5, 5, 8, 16, device=dev)
    conv.bwd_filter(big.ptr(), small.ptr(), dw, 2)
    assert np.abs(dw.cpu().numpy() - z['dK']).max() <= tol * np.abs(z['dK']).max()
