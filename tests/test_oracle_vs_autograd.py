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


@pytest.mark.parametrize('name', ['adagrad', 'adadelta', 'rmsprop_centered', 'momentum'])
def test_optimizer_restatements_match_torch_optim(name):
    """The optimizer restatements (util.py:150-183 branches) against torch.optim's independent statements of the same
    published updates, float64, 6 steps.  TF-specific initial slots (accumulator 0.1, rms 1) are written into torch's
    state; for RMSProp torch keeps eps outside the sqrt, so eps is set to 0 on both sides."""
    rng = np.random.default_rng(5)
    n = 257
    p0 = rng.standard_normal(n)
    grads = [rng.standard_normal(n) for _ in range(6)]
    tp = torch.tensor(p0, dtype=torch.float64, requires_grad=True)
    if name == 'adagrad':
        ref, opt = T.Adagrad(1e-2), torch.optim.Adagrad([tp], lr=1e-2, initial_accumulator_value=0.1, eps=0.0)
    elif name == 'adadelta':
        ref, opt = T.Adadelta(0.5), torch.optim.Adadelta([tp], lr=0.5, rho=0.95, eps=1e-8)
    elif name == 'momentum':
        ref, opt = T.Momentum(1e-2, 0.3), torch.optim.SGD([tp], lr=1e-2, momentum=0.3)
    else:
        ref = T.RMSProp(1e-2, 0.9, 0.3, eps=0.0, centered=True)
        opt = torch.optim.RMSprop([tp], lr=1e-2, alpha=0.9, momentum=0.3, eps=0.0, centered=True)
    P = {'p': p0.copy()}
    for i, g in enumerate(grads):
        tp.grad = torch.tensor(g)
        if name == 'rmsprop_centered' and i == 0:
            opt.step()                                    # materialise the state, then rewind to TF's initial slots
            with torch.no_grad():
                tp.copy_(torch.tensor(p0))
            st = opt.state[tp]
            st['square_avg'].fill_(1.0); st['grad_avg'].zero_(); st['momentum_buffer'].zero_()
        opt.step()
        ref.apply(P, {'p': g})
    assert np.allclose(P['p'], tp.detach().numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('case', [(2, 65, 65, 3, 8, 5, 2), (2, 14, 14, 6, 10, 5, 2), (2, 10, 7, 4, 6, 4, 3), (2, 9, 12, 3, 5, 3, 1)])
def test_valid_padding_conv_family_matches_torch(case):
    """padding='VALID' of the numpy conv family (gen-2 stacks 65 -> 31 -> 14 -> 5 -> 1,
    hem/models/paper_cgan.py:221-224) against torch's conv2d and its autograd, float64."""
    n, h, w, cin, cout, k, s = case
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, h, w, cin))
    Wt = rng.standard_normal((k, k, cin, cout))
    y = T.conv2d(x, Wt, s, 'VALID')
    assert y.shape[1:3] == (T.valid_out(h, k, s), T.valid_out(w, k, s))
    tx = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    tw = torch.tensor(Wt).permute(3, 2, 0, 1).requires_grad_(True)
    ty = torch.nn.functional.conv2d(tx, tw, stride=s)
    assert np.allclose(y, ty.detach().permute(0, 2, 3, 1).numpy(), atol=1e-10)
    dy = rng.standard_normal(y.shape)
    ty.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    dx = T.conv2d_backprop_input(x.shape, Wt, dy, s, 'VALID')
    dw = T.conv2d_backprop_filter(x, Wt.shape, dy, s, 'VALID')
    assert np.allclose(dx, tx.grad.permute(0, 2, 3, 1).numpy(), atol=1e-10)
    assert np.allclose(dw, tw.grad.permute(2, 3, 1, 0).numpy(), atol=1e-10)


def test_forced_derivative_masks_hook_of_the_torch_oracle():
    """oracle/torch_ref.py MASKS (the hook behind the headline-size parity test): with the oracle's OWN masks forced, losses and
    gradients are unchanged to the last bit of float64; with ONE mask entry flipped the loss VALUES are still unchanged (the
    activation's value is untouched) and the gradients move -- i.e. the hook changes derivatives and nothing else."""
    import torch
    from oracle import gan_ref as G
    from oracle import torch_ref as TR
    cfg = G.make_cfg('iwgan', (32, 32, 3), 8, 4)
    P = TR.to_torch(G.init_params(cfg, 3, np.float64), torch.float64)
    rng = np.random.default_rng(2)
    x = torch.tensor(rng.uniform(-1, 1, (4, 3072)))
    z = torch.tensor(rng.standard_normal((4, 8)))
    al = torch.tensor(rng.uniform(0, 1, (4, 1)))

    def run():
        gl, dl = TR.losses(P, x, z, al, cfg)
        gg = TR.grads_of(gl, P, 'generator/')
        dg = TR.grads_of(dl, P, 'discriminator/')
        return float(gl.detach()), float(dl.detach()), {k: v.detach().numpy().copy() for k, v in list(gg.items()) + list(dg.items())}
    assert TR.MASKS is None
    gl0, dl0, g0 = run()
    # the oracle's own masks, recomputed layer by layer
    masks = {}
    with torch.no_grad():
        g_ = 'generator/vars/'
        h = z @ P[g_ + 'fc1/weights'] + P[g_ + 'fc1/bias']
        pre = TR.batch_norm(h, P[G.g_bn_name(0)])
        masks[('g', 0)] = (pre > 0).numpy()
        h = torch.relu(pre).reshape(-1, cfg.s0h, cfg.s0w, 4 * cfg.L)
        for i, name in enumerate(['dc1', 'dc2', 'dc3'], start=1):
            pre = TR.batch_norm(TR.conv2d_transpose_same(h, P[g_ + name + '/weights']) + P[g_ + name + '/bias'], P[G.g_bn_name(i)])
            masks[('g', i)] = (pre > 0).numpy()
            h = torch.relu(pre)
        img = torch.tanh(TR.conv2d_transpose_same(h, P[g_ + 'dc4/weights']) + P[g_ + 'dc4/bias']).reshape(4, -1)
        d_ = 'discriminator/vars/'
        for tag, inp in (('real', x), ('fake', img), ('hat', x + al * (img - x))):
            h = inp.reshape(-1, cfg.H, cfg.W, cfg.C)
            for i, name in enumerate(['c1', 'c2', 'c3']):
                pre = TR.conv2d_same(h, P[d_ + name + '/weights'], 2) + P[d_ + name + '/bias']
                masks[(tag, i)] = (pre > 0).numpy()
                h = TR.lrelu(pre)
    TR.MASKS = masks
    try:
        gl1, dl1, g1 = run()
        assert (gl1, dl1) == (gl0, dl0)
        for k in g0:
            assert np.abs(g1[k] - g0[k]).max() <= 1e-12 * max(1.0, np.abs(g0[k]).max()), k
        flipped = {k: v.copy() for k, v in masks.items()}
        flipped[('hat', 1)].reshape(-1)[5] ^= True
        TR.MASKS = flipped
        gl2, dl2, g2 = run()
        assert gl2 == gl0                                          # the generator loss does not see the x_hat path
        assert abs(dl2 - dl0) <= 1e-9 * max(1.0, abs(dl0)) or dl2 != dl0      # (the penalty's VALUE depends on v = a derivative: it may move)
        k = 'discriminator/vars/c2/weights'
        assert np.abs(g2[k] - g0[k]).max() > 0                     # the critic's gradient moved: the hook reaches the double backward
        assert all(np.array_equal(g2[n], g0[n]) for n in g0 if n.startswith('generator/'))
    finally:
        TR.MASKS = None
