"""GPU parity of the pix2pix cGAN (hem/models/pix2pix.py semantics) against the torch-autograd oracle:
D-step gradients, G-step gradients through the zero-copy skip concats, post-step variables, reported losses."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import pix2pix_ref as PR
from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu


class PairSource:
    def __init__(self, pairs, device):
        self.pairs, self.device, self.i = pairs, device, 0

    def next_batch(self):
        x, y = self.pairs[self.i % len(self.pairs)]
        self.i += 1
        return torch.tensor(x, device=self.device), torch.tensor(y, device=self.device)


def relerr(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def l2err(a, b):
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(np.asarray(a, np.float64).ravel() - b.ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


@pytest.mark.parametrize('variant', ['plain_l1', 'bn_everywhere'])
def test_pix2pix_steps_f32(variant):
    """The 16-layer generator ends in eight batch norms that amplify float32 rounding (torch's own float32
    evaluation of G(x) differs from float64 by 2e-5), and D's gradients at the N(0, 0.02) initial state react to
    such a perturbation of G(x) with ~1e-3 relative changes (flipped lrelu units).  An end-to-end comparison would
    measure that conditioning, not the kernels, so each network is checked with the OTHER network's contribution
    taken from the oracle: (1) G forward, (2) D step on the oracle's G(x), (3) dL/dG(x) through D, (4) U-Net
    backward from the oracle's dL/dG(x) -- norm-wise 1e-3, because a flipped relu unit moves isolated entries --
    and (5) the whole train() policy with the losses of the third batch."""
    p2p, rt, K = pkg('models.pix2pix'), pkg('runtime'), pkg('kernels')
    dev = torch.device('cuda:0')
    bn = variant == 'bn_everywhere'
    B = 2 if bn else 1                      # batch statistics of the 1x1 bottleneck need more than one sample
    args = SimpleNamespace(model='pix2pix', batch_size=B, n_gpus=1, optimizer='rmsprop', lr=1e-4, decay=0.9, momentum=0.01,
                           centered=False, beta1=0.5, beta2=0.999, n_disc_train=1, add_l1=not bn, batch_norm_gen=bn,
                           batch_norm_disc=bn, dropout=0, noise=[])
    P0 = PR.init_params(args, 0, np.float32)
    rng = np.random.default_rng(3)
    pairs = [(rng.uniform(0, 1, (B, 256, 256, 3)).astype(np.float32), rng.uniform(0.01, 0.99, (B, 256, 256, 1)).astype(np.float32))
             for _ in range(3)]
    sess = rt.Session(device=dev, dtype=K.F32, seed=0, rank=0, world_size=1)
    model = p2p.pix2pix(PairSource(pairs, dev), args, sess)
    assert set(model.g_store.index) | set(model.d_store.index) == set(P0)
    model.load_variables(P0)
    P = TR.to_torch(P0, torch.float64)
    x01, y01 = (torch.tensor(v, dtype=torch.float64) for v in pairs[0])
    x, y = 2 * x01 - 1, 2 * y01 - 1

    # (1) generator forward
    g = PR.generator(P, x, args)
    model._load((torch.tensor(pairs[0][0], device=dev), torch.tensor(pairs[0][1], device=dev)))
    model.U.forward()
    slot1 = model.D.x.view(B, B).buf[:B * 256 * 256 * 8].view(B, 256, 256, 8)
    assert np.abs(slot1[..., 3].cpu().numpy() - g.detach().numpy()[..., 0]).max() < 2e-4

    # (2) D step on the oracle's G(x)
    gd = g.detach().clone().requires_grad_(True)
    zr, zf = PR.discriminator(P, x, y, args, 0), PR.discriminator(P, x, gd, args, 1)
    d_total = PR.xent(zr, 1.0).mean() + PR.xent(zf, 0.0).mean()
    ref = TR.grads_of(d_total, P, 'discriminator/')
    slot1[..., 3] = torch.tensor(g.detach().numpy()[..., 0], dtype=torch.float32, device=dev)
    model._d_forward(0, 2)
    model._xent(1)
    if bn:
        model.D.backward(0, B, bn_pass=0, want_params=True, acc=False)
        model.D.backward(B, B, bn_pass=1, want_params=True, acc=True)
    else:
        model.D.backward(0, 2 * B, want_params=True)
    got = model.gradients()
    for k, v in ref.items():
        if bn and k.endswith('/bias') and k[-7:-5] in ('m2', 'm3', 'm4', 'm5'):
            continue                        # biases feeding batch norm: zero gradient up to rounding
        assert relerr(got[k], v.numpy()) < 1e-3, k

    # (3) dL/dG(x) of the generator loss through D (+ the L1 term)
    g_total = PR.xent(zf, 1.0).mean()
    if args.add_l1:
        g_total = g_total + 10.0 * ((y + 1) / 2 - (gd + 1) / 2).abs().mean()
    seed = torch.autograd.grad(g_total, gd, retain_graph=True)[0]
    model._xent(2)
    model.D.backward(B, B, bn_pass=1, want_params=False, want_dx=True)
    if args.add_l1:
        model._l1(True)
    dx1 = model.D.dx.view(B, B).buf[:B * 256 * 256 * 8].view(B, 256, 256, 8)
    assert relerr(dx1[..., 3].cpu().numpy(), seed.numpy()[..., 0]) < 1e-3

    # (4) U-Net backward from the oracle's dL/dG(x).  Batch-norm backward over the 4..16 samples of the bottleneck
    # layers cancels heavily, so even torch's float32 evaluation of these gradients is ~1e-2 away from float64: the bar
    # per tensor is max(3e-3, 10 x that measured float32 sensitivity of the oracle) in the max norm (which a few elements
    # decide: another float32 summation order moves it by small factors) and max(1e-3, 10 x) in the l2 norm
    # (the yardstick itself moves by ~2x with the host's thread count: 5 x failed once at 0.0081 vs 0.0070 on one box).
    gkeys = [k for k in P if k.startswith('generator/')]
    ref = dict(zip(gkeys, torch.autograd.grad(g, [P[k] for k in gkeys], grad_outputs=seed)))
    P32 = TR.to_torch(P0, torch.float32)
    g32 = PR.generator(P32, x.float(), args)
    ref32 = dict(zip(gkeys, torch.autograd.grad(g32, [P32[k] for k in gkeys], grad_outputs=seed.float())))
    dx1[..., 3] = torch.tensor(seed.numpy()[..., 0], dtype=torch.float32, device=dev)
    model.U.backward()
    got = model.gradients()
    for k, v in ref.items():
        if k.endswith('/bias') and ('decoder' in k or (bn and not k.endswith('/1/bias'))):
            continue
        tol = max(3e-3, 10.0 * relerr(ref32[k].double().numpy(), v.numpy()))
        tol2 = max(1e-3, 10.0 * l2err(ref32[k].double().numpy(), v.numpy()))
        assert relerr(got[k], v.numpy()) < tol, (k, tol, l2err(got[k], v.numpy()), relerr(got[k], v.numpy()))
        assert l2err(got[k], v.numpy()) < tol2, (k, tol2, l2err(got[k], v.numpy()))

    # (5) the training policy end to end: losses of the third batch after one D and one G step
    model.load_variables(P0)
    model.x_y.i = 0
    out = model.train(sess, args, None)
    tr = PR.Trainer(TR.to_torch(P0, torch.float64), args)
    rep = tr.train([(torch.tensor(a, dtype=torch.float64), torch.tensor(b, dtype=torch.float64)) for a, b in pairs])
    assert set(out) == set(rep), (set(out), set(rep))
    for k in rep:
        assert abs(out[k] - rep[k]) < 2e-3 * max(1.0, abs(rep[k])), (k, out[k], rep[k])


def test_pix2pix_bf16_train_runs():
    p2p, rt, K = pkg('models.pix2pix'), pkg('runtime'), pkg('kernels')
    dev = torch.device('cuda:0')
    args = SimpleNamespace(model='pix2pix', batch_size=2, n_gpus=1, optimizer='adam', lr=1e-4, beta1=0.5, beta2=0.999,
                           decay=0.9, momentum=0.01, centered=False, n_disc_train=1, add_l1=True)
    rng = np.random.default_rng(0)
    pairs = [(rng.uniform(0, 1, (2, 256, 256, 3)).astype(np.float32), rng.uniform(0.01, 0.99, (2, 256, 256, 1)).astype(np.float32))
             for _ in range(3)]
    sess = rt.Session(device=dev, dtype=K.BF16, seed=0, rank=0, world_size=1)
    model = pkg('models').get_model('pix2pix')(PairSource(pairs, dev), args, sess)
    out = model.train(sess, args, None)
    assert set(out) == {'l1', 'add', 'total', 'd_real', 'd_fake', 'rmse'} and all(np.isfinite(v) for v in out.values())


def test_pix2pix_dropout_forward_and_backward_f32():
    """`--dropout 0.5` (examples/pix2pix/baseline.config): tf.nn.dropout(h, keep_prob) on decoder layers 1-3 with injected
    uniform draws -- G(x) and the U-Net's parameter gradients from an injected dL/dG(x) against the float64 oracle."""
    p2p, rt, K = pkg('models.pix2pix'), pkg('runtime'), pkg('kernels')
    dev = torch.device('cuda:0')
    B, keep = 2, 0.5
    args = SimpleNamespace(model='pix2pix', batch_size=B, n_gpus=1, optimizer='rmsprop', lr=1e-4, decay=0.9, momentum=0.01,
                           centered=False, beta1=0.5, beta2=0.999, n_disc_train=1, add_l1=False, batch_norm_gen=True,
                           batch_norm_disc=False, dropout=keep, noise=[])
    P0 = PR.init_params(args, 0, np.float32)
    rng = np.random.default_rng(5)
    x01 = rng.uniform(0, 1, (B, 256, 256, 3)).astype(np.float32)
    y01 = rng.uniform(0.01, 0.99, (B, 256, 256, 1)).astype(np.float32)
    drops = [rng.uniform(0, 1, (B, 2 << i, 2 << i, 512)).astype(np.float32) for i in range(3)]
    seed = rng.standard_normal((B, 256, 256, 1)).astype(np.float32) * 1e-3
    sess = rt.Session(device=dev, dtype=K.F32, seed=0, rank=0, world_size=1)
    model = p2p.pix2pix(PairSource([(x01, y01)], dev), args, sess)
    model.load_variables(P0)
    P = TR.to_torch(P0, torch.float64)
    x = torch.tensor(2 * x01.astype(np.float64) - 1)
    g = PR.generator(P, x, args, [torch.tensor(d, dtype=torch.float64) for d in drops])
    gkeys = [k for k in P if k.startswith('generator/')]
    ref = dict(zip(gkeys, torch.autograd.grad(g, [P[k] for k in gkeys], grad_outputs=torch.tensor(seed, dtype=torch.float64))))
    P32 = TR.to_torch(P0, torch.float32)
    g32 = PR.generator(P32, x.float(), args, [torch.tensor(d) for d in drops])
    ref32 = dict(zip(gkeys, torch.autograd.grad(g32, [P32[k] for k in gkeys], grad_outputs=torch.tensor(seed))))

    model._load((torch.tensor(x01, device=dev), torch.tensor(y01, device=dev)))
    sess.inject = {'dropout': [d for d in drops]}
    model.U.forward()
    assert not sess.inject.get('dropout')                      # all three draws consumed, in layer order
    slot1 = model.D.x.view(B, B).buf[:B * 256 * 256 * 8].view(B, 256, 256, 8)
    assert np.abs(slot1[..., 3].cpu().numpy() - g.detach().numpy()[..., 0]).max() < 5e-4
    dx1 = model.D.dx.view(B, B).buf[:B * 256 * 256 * 8].view(B, 256, 256, 8)
    dx1.zero_()
    dx1[..., 3] = torch.tensor(seed[..., 0], device=dev)
    model.U.backward()
    got = model.gradients()
    for k, v in ref.items():
        if k.endswith('/bias') and ('decoder' in k or not k.endswith('/1/bias')):
            continue                                            # biases feeding batch norm
        # l2 only: with a white-noise dL/dG(x) a float32 difference in a pre-activation near zero flips a relu unit of
        # the 2-sample bottleneck layers and moves isolated filter entries by 10-20 % in ANY float32 implementation
        # (torch's own float32 run: max-norm 5e-2, l2 4e-3 from its float64 run on the same inputs); a wrong mask or a
        # missing 1/keep would be O(1) in l2
        tol2 = max(1e-2, 5.0 * l2err(ref32[k].double().numpy(), v.numpy()))
        assert l2err(got[k], v.numpy()) < tol2, (k, tol2, l2err(got[k], v.numpy()))


@pytest.mark.parametrize('which', [['input'], ['latent', 'end'], ['input', 'latent', 'end']])
def test_pix2pix_noise_injection_f32(which):
    """`--noise input|latent|end` (hem/models/pix2pix.py:183-186,204-206,223-225; examples/pix2pix/noise*.config): a
    U(-1,1) tensor concatenated to the generator input (4 -> 64), to the 1x1 bottleneck (1024 -> 512) and to the last
    decoder layer's input (129 -> 1).  With the uniform draws injected on both sides: G(x), and the U-Net's parameter
    gradients from an injected dL/dG(x), against the float64 oracle -- including the filter slices that see only noise."""
    p2p, rt, K = pkg('models.pix2pix'), pkg('runtime'), pkg('kernels')
    dev = torch.device('cuda:0')
    B = 2
    args = SimpleNamespace(model='pix2pix', batch_size=B, n_gpus=1, optimizer='rmsprop', lr=1e-4, decay=0.9, momentum=0.01,
                           centered=False, beta1=0.5, beta2=0.999, n_disc_train=1, add_l1=False, batch_norm_gen=False,
                           batch_norm_disc=False, dropout=0, noise=list(which))
    P0 = PR.init_params(args, 0, np.float32)
    shapes = {'input': (B, 256, 256, 1), 'latent': (B, 1, 1, 512), 'end': (B, 128, 128, 1)}
    for k, want in (('generator/enocder/vars/1/weights', (4, 4, 4 if 'input' in which else 3, 64)),
                    ('generator/decoder/vars/1/weights', (4, 4, 512, 1024 if 'latent' in which else 512)),
                    ('generator/decoder/vars/8/weights', (4, 4, 1, 129 if 'end' in which else 128))):
        assert P0[k].shape == want
    rng = np.random.default_rng(8)
    x01 = rng.uniform(0, 1, (B, 256, 256, 3)).astype(np.float32)
    y01 = rng.uniform(0.01, 0.99, (B, 256, 256, 1)).astype(np.float32)
    u01 = {k: rng.uniform(0, 1, shapes[k]).astype(np.float32) for k in which}          # the draws in [0, 1) as the device sees them
    seed = rng.standard_normal((B, 256, 256, 1)).astype(np.float32) * 1e-3
    sess = rt.Session(device=dev, dtype=K.F32, seed=0, rank=0, world_size=1)
    model = p2p.pix2pix(PairSource([(x01, y01)], dev), args, sess)
    assert set(model.g_store.index) | set(model.d_store.index) == set(P0)
    model.load_variables(P0)
    P = TR.to_torch(P0, torch.float64)
    x = torch.tensor(2 * x01.astype(np.float64) - 1)
    noise = {k: torch.tensor(2.0 * (u.astype(np.float64) - 0.5)) for k, u in u01.items()}
    g = PR.generator(P, x, args, noise=noise)
    gkeys = [k for k in P if k.startswith('generator/')]
    ref = dict(zip(gkeys, torch.autograd.grad(g, [P[k] for k in gkeys], grad_outputs=torch.tensor(seed, dtype=torch.float64))))
    P32 = TR.to_torch(P0, torch.float32)
    g32 = PR.generator(P32, x.float(), args, noise={k: v.float() for k, v in noise.items()})
    ref32 = dict(zip(gkeys, torch.autograd.grad(g32, [P32[k] for k in gkeys], grad_outputs=torch.tensor(seed))))

    model._load((torch.tensor(x01, device=dev), torch.tensor(y01, device=dev)))
    sess.inject = {'noise_' + k: [u] for k, u in u01.items()}
    model.U.forward()
    assert not any(sess.inject.values())                       # every requested draw consumed
    slot1 = model.D.x.view(B, B).buf[:B * 256 * 256 * 8].view(B, 256, 256, 8)
    assert np.abs(slot1[..., 3].cpu().numpy() - g.detach().numpy()[..., 0]).max() < 5e-4
    dx1 = model.D.dx.view(B, B).buf[:B * 256 * 256 * 8].view(B, 256, 256, 8)
    dx1.zero_()
    dx1[..., 3] = torch.tensor(seed[..., 0], device=dev)
    model.U.backward()
    got = model.gradients()
    for k, v in ref.items():
        if k.endswith('/bias') and 'decoder' in k:
            continue                                            # biases feeding batch norm
        tol2 = max(1e-2, 5.0 * l2err(ref32[k].double().numpy(), v.numpy()))
        assert l2err(got[k], v.numpy()) < tol2, (k, tol2, l2err(got[k], v.numpy()))
    # a free-running train() with device-drawn noise: finite losses, fresh noise per generator pass (3 passes x |which| draws)
    sess.inject = {}
    d0 = sess.rng_state()
    out = model.train(sess, args, None)
    assert all(np.isfinite(v) for v in out.values())
    assert sess.rng_state() - d0 == 3 * len(which)
