"""Closed-form answers on the HIP path itself (no oracle in the loop): the analytic anchors SURVEY.md section 8c lists for a
build whose oracle cannot be pinned to TensorFlow.  Each holds at any size, so they also run at the benchmark's batch."""
import math

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T
from test_gpu_gan_step import ListSource, make_args

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,L', [(8, 8), (512, 200)])
def test_gradient_penalty_of_a_linear_critic(B, L):
    """models/gan.py:224-230 with a critic made linear in its input (positive weights, large positive biases: every lrelu unit
    is active): d D(x_hat) / d x_hat is the SAME vector w_eff for every row, so slopes = sqrt(sum over the whole batch) =
    sqrt(B) * |w_eff| (the reference's whole-batch norm; a per-sample norm would give |w_eff|) and
    penalty = (sqrt(B) |w_eff| - 1)^2, independent of x, g and alpha."""
    gan, rt, K = pkg('models.gan'), pkg('runtime'), pkg('kernels')
    dev = torch.device('cuda:0')
    shape = (32, 32, 3)
    args = make_args('iwgan', B, L, shape)
    rng = np.random.default_rng(5)
    sess = rt.Session(device=dev, dtype=K.F32, seed=1, rank=0, world_size=1)
    rep = gan.GanReplica(ListSource([rng.uniform(0, 1, (B,) + shape).astype(np.float32)], dev), args, sess)
    P = rep.variables()
    for k in P:
        if k.startswith('discriminator'):
            P[k] = (np.abs(rng.standard_normal(P[k].shape)) * (0.02 if k.endswith('weights') else 0.0) +
                    (0.0 if k.endswith('weights') else 50.0)).astype(np.float32)
    rep.load_variables(P)
    rep.use_graphs = False
    rep.d_step(rep.x_source.next_batch())
    torch.cuda.synchronize()
    v = rep.D.dx.get()[2 * B:3 * B].reshape(B, -1).astype(np.float64)          # d D(x_hat) / d x_hat, slot 2
    assert np.abs(v - v[0]).max() < 1e-5 * np.abs(v[0]).max()                  # one w_eff for every row
    w = np.linalg.norm(v[0])
    s = rep.scal.cpu().numpy().astype(np.float64)
    slopes_sq = s[rep.S_SUMSQ]
    assert abs(slopes_sq - B * w * w) < 1e-4 * B * w * w
    assert abs(s[rep.S_GP] - (math.sqrt(B) * w - 1.0) ** 2) < 1e-4 * max(1.0, (math.sqrt(B) * w - 1.0) ** 2)
    assert abs(s[rep.S_GP] - (w - 1.0) ** 2) > 1e-3                            # not the per-sample form


@pytest.mark.parametrize('dtype', [0, 1])
def test_identity_filters_and_shapes(dtype):
    """A 1x1 identity filter returns its input; a centre-tap 5x5 stride-1 identity too; conv2d SAME halves (ceil) and
    conv2d_transpose doubles the spatial size (ops/layers.py:101,140-142)."""
    K = pkg('kernels')
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(2)
    n, h, w, c = 3, 9, 12, 16
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    for k in (1, 5):
        big, small = K.Act(n, h, w, c, dtype, dev), K.Act(n, h, w, c, dtype, dev)
        conv = K.Conv(big, small, k, k, 1, k // 2, k // 2)
        Wt = np.zeros((k, k, c, c), np.float32)
        Wt[k // 2, k // 2] = np.eye(c)
        conv.pack(torch.tensor(Wt, device=dev))
        big.set(x)
        conv.fwd(big.ptr(), small.ptr(), n)
        assert np.array_equal(small.get(), big.get())
        conv.bwd_data(small.ptr(), big.like().ptr(), n)                      # (runs; its value is the adjoint test's business)
    oh, pt, _ = T.same_pad(h, 5, 2)
    ow, pl, _ = T.same_pad(w, 5, 2)
    assert (oh, ow) == (5, 6)                                                # ceil(in / stride)
    big, small = K.Act(n, 2 * oh, 2 * ow, 8, dtype, dev), K.Act(n, oh, ow, c, dtype, dev)
    conv = K.Conv(big, small, 5, 5, 2, *[T.same_pad(d, 5, 2)[1] for d in (2 * oh, 2 * ow)])
    conv.pack(torch.tensor(rng.standard_normal((5, 5, 8, c)).astype(np.float32), device=dev))
    small.set(rng.standard_normal((n, oh, ow, c)).astype(np.float32))
    conv.bwd_data(small.ptr(), big.ptr(), n)                                 # conv2d_transpose: [n, 2 oh, 2 ow, 8]
    assert np.isfinite(big.get()).all() and np.abs(big.get()).max() > 0


@pytest.mark.parametrize('dtype', [0, 1])
def test_batch_norm_moments_and_lrelu_points(dtype):
    """contrib batch_norm in training mode (scale=False, eps 1e-3, biased variance): per channel the output has mean beta and
    variance var / (var + eps); lrelu(+-1) = (1, -0.2)."""
    K, lib = pkg('kernels'), pkg('_lib')
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(3)
    rows, c = 4096, 24
    u = (rng.standard_normal((rows, c)) * rng.uniform(0.5, 3, c) + rng.uniform(-2, 2, c)).astype(np.float32)
    ua = K.Act(rows, 1, 1, c, dtype, dev).set(u)
    pre, h = ua.like(), ua.like()
    beta = torch.tensor(rng.standard_normal(c).astype(np.float32), device=dev)
    stats = torch.zeros(2 * c, device=dev)
    K.bn_fwd(K.Workspace(dev), ua, c, beta, K.ACT_NONE, pre, h, stats)
    y = pre.get().reshape(rows, c).astype(np.float64)
    var = ua.get().reshape(rows, c).astype(np.float64).var(0)
    tol = 1e-4 if dtype == 0 else 2e-2
    assert np.abs(y.mean(0) - beta.cpu().numpy()).max() < tol
    assert np.abs(y.var(0) - var / (var + 1e-3)).max() < tol
    pts = K.Act(2, 1, 1, 8, dtype, dev).set(np.stack([np.ones((1, 1, 8)), -np.ones((1, 1, 8))]).astype(np.float32))
    out = pts.like()
    lib.call('tdg_bias_act', dtype, pts.ptr(), 2, 8, pts.cs, None, K.ACT_LRELU, 0.2, out.ptr(), K.stream())
    got = out.get().reshape(2, 8)
    assert np.allclose(got[0], 1.0) and np.allclose(got[1], -0.2, atol=1e-3)
