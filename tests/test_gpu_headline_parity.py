"""Parity of the path that bench.py TIMES, at the size it is timed at (BASELINE.json configs[1]:
`--model iwgan --batch_size 512 --latent_size 200`, 32x32x3, adam 1e-4 / 0.5 / 0.9 = examples/iwgan.config).

(a) kernel level: the exact headline conv launches -- c2 (16x16x200 -> 8x8x400) and c3 (8x8x400 -> 4x4x800) on the
    critic's batched 3B = 1536 images, bf16, DEFAULT tile / split selection (no TDG_* override) -- forward,
    backward-data, filter gradient and the two-source filter gradient (n_first = 1024) against the float64 NumPy
    oracle on bf16-rounded inputs.  Bound: 2e-2 of the output's max magnitude (bf16 operands, f32 accumulation, bf16
    store for activations / f32 store for filter gradients).
(b) step level, f32: one D step + one G step at B = 512, L = 200 with injected z / alpha against the float64
    torch-autograd oracle (oracle/torch_ref.py, `create_graph=True` for the penalty).  Bound: the north-star's 1e-3
    on the losses and on every gradient (relative to the tensor's max magnitude).
(c) step level, bf16 (the timed dtype) against the HIP f32 run of (b) on identical inputs.  Bound per tensor:
    relative l2 error <= BF16_REL_L2 for every weight gradient, losses within BF16_LOSS_TOL.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T
from oracle import gan_ref as G
from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu

N_HEADLINE = 1536                  # 3 x 512: [x | g | x_hat] rows of one batched critic pass
BF16_TOL = 2e-2
BF16_REL_L2 = 3e-2                 # per-tensor |g_bf16 - g_f32|_2 / |g_f32|_2 of a whole D / G step (measured: see DESIGN.md s.2)
BF16_LOSS_TOL = 2e-2


def bf16_round(a):
    return torch.tensor(a, dtype=torch.float32).bfloat16().float().numpy()


def relerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def last_kernel():
    return pkg('_lib').load().tdg_last_kernel().decode()


HEADLINE_CONVS = [
    # name, h, w, cin, cout  (k = 5, stride 2: models/gan.py:281-282 at L = 200)
    ('c2', 16, 16, 200, 400),
    ('c3', 8, 8, 400, 800),
]


@pytest.mark.parametrize('case', HEADLINE_CONVS, ids=lambda c: c[0])
def test_headline_conv_launches_bf16(case):
    K = pkg('kernels')
    name, h, w, cin, cout = case
    n, k, s = N_HEADLINE, 5, 2
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(len(name) + cin)
    oh, pt, _ = T.same_pad(h, k, s)
    ow, pl, _ = T.same_pad(w, k, s)
    big = K.Act(n, h, w, cin, K.BF16, dev)
    small = K.Act(n, oh, ow, cout, K.BF16, dev)
    conv = K.Conv(big, small, k, k, s, pt, pl)
    x = bf16_round(rng.standard_normal((n, h, w, cin)).astype(np.float32))
    Wt = bf16_round((rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32))
    b = rng.standard_normal(cout).astype(np.float32)
    dy = bf16_round(rng.standard_normal((n, oh, ow, cout)).astype(np.float32))
    conv.pack(torch.tensor(Wt, device=dev))
    W64 = Wt.astype(np.float64)
    # images checked against the oracle for the per-image forms: both ends, the slot boundaries of the 3B batch and
    # the rows around every 192 / 256-row tile seam are inside these images' pixels
    idx = np.unique(np.concatenate([[0, 1, 2, 511, 512, 513, 1023, 1024, 1025, n - 2, n - 1],
                                    rng.integers(0, n, 24)]))
    # ---- forward + bias + lrelu (the critic's layer as it runs in the step)
    big.set(x)
    conv.fwd(big.ptr(), small.ptr(), n, K.epilogue(bias=torch.tensor(b, device=dev), act=K.ACT_LRELU, leak=0.2))
    kern = last_kernel()
    assert 'igemm_fwd_dma_kernel<bf16' in kern and ',208,' in kern, kern
    got = small.get()
    ref = T.lrelu(T.conv2d(x[idx].astype(np.float64), W64, s) + b)
    assert relerr(got[idx], ref) < BF16_TOL, kern
    assert np.isfinite(got).all()
    # every image, cheaply: column sums of the output against the oracle's linearity (sum over images of the
    # PRE-activation is the conv of the image sum) is not available behind the lrelu, so check a second random subset
    idx2 = rng.integers(0, n, 16)
    assert relerr(got[idx2], T.lrelu(T.conv2d(x[idx2].astype(np.float64), W64, s) + b)) < BF16_TOL
    # ---- backward-data with the lrelu mask of the layer below
    small.set(dy)
    mask = big.like().set(x)
    out = big.like()
    conv.bwd_data(small.ptr(), out.ptr(), n, K.epilogue(mask_mode=K.MASK_LRELU, mask_src=mask.ptr(), leak=0.2))
    kern = last_kernel()
    assert 'igemm_fwd_dma_kernel<bf16' in kern and ',208,' in kern, kern
    got = out.get()
    ref = T.conv2d_backprop_input((len(idx), h, w, cin), W64, dy[idx].astype(np.float64), s) * \
        T.lrelu_grad_mask(x[idx].astype(np.float64))
    assert relerr(got[idx], ref) < BF16_TOL, kern
    # ---- filter gradient over all 1536 images (default split count), oracle accumulated over image chunks
    dw = torch.zeros(k, k, cin, cout, device=dev)
    conv.bwd_filter(big.ptr(), small.ptr(), dw, n)
    ref = np.zeros((k, k, cin, cout))
    for i0 in range(0, n, 128):
        ref += T.conv2d_backprop_filter(x[i0:i0 + 128].astype(np.float64), Wt.shape, dy[i0:i0 + 128].astype(np.float64), s)
    assert relerr(dw.cpu().numpy(), ref) < BF16_TOL
    # ---- two-source filter gradient as the D step issues it: rows of [h(x) | h(g)] (1024 images) from one tensor,
    # the 512 tangent rows from another, against ONE delta tensor (engine.SeqNet.merged_wgrad)
    n_first = 1024
    x2 = bf16_round(rng.standard_normal((n - n_first, h, w, cin)).astype(np.float32))
    tan = K.Act(n - n_first, h, w, cin, K.BF16, dev).set(x2)
    dw2 = torch.full((k, k, cin, cout), 0.25, device=dev)
    conv.bwd_filter2(big.ptr(), n_first, tan.ptr(), small.ptr(), dw2, n, beta=1.0)
    ref2 = ref.copy() + 0.25
    for i0 in range(n_first, n, 128):
        ref2 -= T.conv2d_backprop_filter(x[i0:i0 + 128].astype(np.float64), Wt.shape, dy[i0:i0 + 128].astype(np.float64), s)
        ref2 += T.conv2d_backprop_filter(x2[i0 - n_first:i0 - n_first + 128].astype(np.float64), Wt.shape,
                                         dy[i0:i0 + 128].astype(np.float64), s)
    assert relerr(dw2.cpu().numpy(), ref2) < BF16_TOL


# ------------------------------------------------------------------------------------------------ whole steps
B, L, SHAPE = 512, 200, (32, 32, 3)


def headline_args():
    return SimpleNamespace(model='iwgan', batch_size=B, latent_size=L, image_shape=SHAPE, n_gpus=1, optimizer='adam',
                           lr=1e-4, beta1=0.5, beta2=0.9, decay=0.9, momentum=0.01, centered=False, n_disc_train=5,
                           display_d_loss=True, use_graphs=False)


class _Batches:
    def __init__(self, batches, device):
        self.batches, self.device, self.i = batches, device, 0

    def next_batch(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return torch.tensor(b, dtype=torch.float32, device=self.device)


def _inputs():
    rng = np.random.default_rng(2024)
    xs = [(rng.integers(0, 256, (B,) + SHAPE).astype(np.float32) / 255.0) for _ in range(2)]     # CIFAR-shaped bytes
    zs = [rng.standard_normal((B, L)).astype(np.float32) for _ in range(2)]
    als = [rng.uniform(0, 1, (B, 1)).astype(np.float32) for _ in range(2)]
    return xs, zs, als


def _hip_d_then_g(dtype, P, xs, zs, als):
    """One D step then one G step on the HIP path; returns (D grads, d scalars, G grads, G-step losses)."""
    gan, rt = pkg('models.gan'), pkg('runtime')
    dev = torch.device('cuda:0')
    sess = rt.Session(device=dev, dtype=dtype, seed=0, rank=0, world_size=1)
    rep = gan.GanReplica(_Batches(xs, dev), headline_args(), sess)
    rep.load_variables({k: np.asarray(v, np.float32) for k, v in P.items()})
    sess.inject = {'z': [zs[0]], 'alpha': [als[0]]}
    rep.d_step(rep.x_source.next_batch())
    dgr = {k: v for k, v in rep.gradients().items() if k.startswith('discriminator/')}
    s = rep.scal.cpu().numpy().astype(np.float64)
    d_loss = s[rep.S_DFAKE] - s[rep.S_DREAL] + 10.0 * s[rep.S_GP]
    sess.inject = {'z': [zs[1]], 'alpha': [als[1]]}
    rep.g_step(rep.x_source.next_batch())
    ggr = {k: v for k, v in rep.gradients().items() if k.startswith('generator/')}
    out = rep.losses()
    del rep
    torch.cuda.empty_cache()
    return dgr, d_loss, ggr, out


_cache = {}


def _f32_run():
    if 'f32' not in _cache:
        cfg = G.make_cfg('iwgan', SHAPE, L, B)
        P = G.init_params(cfg, 0, np.float64)
        xs, zs, als = _inputs()
        _cache['f32'] = (cfg, P, xs, zs, als, _hip_d_then_g(0, P, xs, zs, als))
    return _cache['f32']


BN_FED = {'generator/vars/%s/bias' % n for n in ('fc1', 'dc1', 'dc2', 'dc3')}       # zero gradient up to rounding


def test_headline_step_f32_vs_float64_autograd_oracle():
    """B = 512, L = 200: the D step's loss and every critic gradient, then (after the oracle's own Adam update of D) the
    G step's losses and every generator gradient, within 1e-3 of the float64 torch-autograd oracle."""
    cfg, P, xs, zs, als, (dgr, d_loss, ggr, out) = _f32_run()
    torch.set_num_threads(max(1, min(16, len(__import__('os').sched_getaffinity(0)))))
    P64 = TR.to_torch(P, torch.float64)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    _, dl = TR.losses(P64, TR.TorchGanTrainer.rescale(t(xs[0])), t(zs[0]), t(als[0]), cfg)
    ref = TR.grads_of(dl, P64, 'discriminator/')
    assert abs(d_loss - float(dl)) < 1e-3 * max(1.0, abs(float(dl))), (d_loss, float(dl))
    worst = {}
    for k, g in ref.items():
        worst[k] = relerr(dgr[k], g.detach().numpy())
        assert worst[k] < 1e-3, (k, worst[k])
    # the oracle's D update, then the G step on the updated critic
    opt = TR.TorchAdam(1e-4, 0.5, 0.9)
    opt.apply(P64, ref)
    gl, dl2 = TR.losses(P64, TR.TorchGanTrainer.rescale(t(xs[1])), t(zs[1]), t(als[1]), cfg)
    gref = TR.grads_of(gl, P64, 'generator/')
    assert abs(out['g_loss'] - float(gl)) < 1e-3 * max(1.0, abs(float(gl)))
    assert abs(out['d_loss'] - float(dl2)) < 1e-3 * max(1.0, abs(float(dl2)))
    for k, g in gref.items():
        if k in BN_FED:
            continue
        worst[k] = relerr(ggr[k], g.detach().numpy())
        assert worst[k] < 1e-3, (k, worst[k])
    print('headline f32 step vs f64 oracle, max |err| / max |ref| per tensor: ' +
          ', '.join('%s %.1e' % (k.split('/')[-2] + '.' + k.split('/')[-1][0], v) for k, v in worst.items()))


def test_headline_step_bf16_vs_f32():
    """The timed dtype against the f32 HIP run of the same step (identical weights, batch, z, alpha)."""
    cfg, P, xs, zs, als, (dgr, d_loss, ggr, out) = _f32_run()
    dgr_b, d_loss_b, ggr_b, out_b = _hip_d_then_g(1, P, xs, zs, als)
    assert abs(d_loss_b - d_loss) < BF16_LOSS_TOL * max(1.0, abs(d_loss)), (d_loss_b, d_loss)
    for k in ('g_loss', 'd_loss'):
        assert abs(out_b[k] - out[k]) < BF16_LOSS_TOL * max(1.0, abs(out[k])), (k, out_b[k], out[k])
    worst = {}
    for k, g in list(dgr.items()) + list(ggr.items()):
        if k in BN_FED:
            continue
        gb = (dgr_b if k in dgr_b else ggr_b)[k]
        worst[k] = rel_l2(gb, g)
    print('headline bf16 step vs f32, relative l2 error per tensor: ' +
          ', '.join('%s %.1e' % (k.split('/')[-2] + '.' + k.split('/')[-1][0], v) for k, v in worst.items()))
    for k, v in worst.items():
        if k.endswith('/weights'):
            assert v < BF16_REL_L2, (k, v)
        else:
            assert v < 2 * BF16_REL_L2, (k, v)          # biases / betas: small sums of bf16-rounded deltas
