"""Pins the CPU oracle with every known answer the reference's own tests and source offer
(SURVEY.md section 8c): hem/ops/test_losses.py:6-27 (rmse), hem/util/test_misc.py:7-30
(collection_to_dict, chunks), plus analytic identities derivable from the reference source."""
import importlib
import math

import numpy as np

from oracle import tf_ops as T
from oracle import gan_ref as G


def test_rmse_reference_vectors():
    """hem/ops/test_losses.py:6-27: rmse(1,1)=0, rmse(1,0)=1, rmse(-1,1)=2, rmse(1,-1)=2 on (1,64,64,3)."""
    ones, zeros = np.ones((1, 64, 64, 3)), np.zeros((1, 64, 64, 3))
    assert np.allclose(T.rmse(ones, ones), 0)
    assert np.allclose(T.rmse(ones, zeros), 1)
    assert np.allclose(T.rmse(-ones, ones), 2)
    assert np.allclose(T.rmse(ones, -ones), 2)


def test_collection_to_dict_and_chunks_reference_vectors():
    """hem/util/test_misc.py:7-30, against the product's util (pure host logic)."""
    util = importlib.import_module('3dgan_amd.util')

    class Named:
        def __init__(self, name):
            self.name = name
    a, b = Named('a:0'), Named('tests/b:0')
    d = util.collection_to_dict([a, b])
    assert d['a'] is a and d['b'] is b
    y = list(util.chunks(list(range(10)), 5))
    assert len(y) == 2 and y[0] == [0, 1, 2, 3, 4] and y[1] == [5, 6, 7, 8, 9]


def test_lrelu_and_gradient_rule():
    """ops/activations.py:28 max(leak*x, x); TF MaximumGrad sends the tie at 0 to leak*x."""
    x = np.array([-1.0, 0.0, 1.0])
    assert np.allclose(T.lrelu(x), [-0.2, 0.0, 1.0])
    assert np.allclose(T.lrelu_grad_mask(x), [0.2, 0.2, 1.0])


def test_same_padding_geometry():
    """App. A-1: k5 s2 on 32/64 -> pad (1,2); k4 s2 on 256 -> (1,1); k5 s2 on 28 -> out 14 pad (1,2)."""
    assert T.same_pad(32, 5, 2) == (16, 1, 2)
    assert T.same_pad(64, 5, 2) == (32, 1, 2)
    assert T.same_pad(256, 4, 2) == (128, 1, 1)
    assert T.same_pad(28, 5, 2) == (14, 1, 2)
    assert T.same_pad(7, 1, 1) == (7, 0, 0)
    assert T.valid_out(65, 5, 2) == 31 and T.valid_out(31, 5, 2) == 14


def test_identity_kernel_conv_and_deconv_shape():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 6, 6, 3))
    K = np.zeros((5, 5, 3, 3))
    K[2, 2] = np.eye(3)                         # centre tap identity, stride 1 -> y == x
    assert np.allclose(T.conv2d(x, K, 1), x)
    # stride-2 SAME with pad_before 1: y[oh,ow] = x[2oh+1, 2ow+1]
    assert np.allclose(T.conv2d(x, K, 2), x[:, 1::2, 1::2, :])
    y = T.conv2d_transpose(x, rng.standard_normal((5, 5, 4, 3)), (2, 12, 12, 4), 2)   # ops/layers.py:140-141
    assert y.shape == (2, 12, 12, 4)


def test_conv_backprops_are_adjoints():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 7, 9, 3))
    K = rng.standard_normal((5, 5, 3, 4))
    y = T.conv2d(x, K, 2)
    dy = rng.standard_normal(y.shape)
    # <conv(x,K), dy> == <x, bwd_input(dy)> == <K, bwd_filter(x,dy)>
    lhs = np.sum(y * dy)
    assert np.allclose(lhs, np.sum(x * T.conv2d_backprop_input(x.shape, K, dy, 2)))
    assert np.allclose(lhs, np.sum(K * T.conv2d_backprop_filter(x, K.shape, dy, 2)))


def test_batch_norm_contract():
    """App. A-3: no gamma, biased variance, eps 1e-3, beta shift; backward kills mean and xhat components."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((8, 4, 4, 5)) * 3 + 1
    beta = rng.standard_normal(5)
    y, cache = T.batch_norm_train(x, beta)
    yc = y - beta
    assert np.allclose(yc.mean(axis=(0, 1, 2)), 0, atol=1e-12)
    var = x.var(axis=(0, 1, 2))
    assert np.allclose(yc.var(axis=(0, 1, 2)), var / (var + 1e-3))
    dy = rng.standard_normal(x.shape)
    dx, dbeta = T.batch_norm_train_backward(dy, cache)
    assert np.allclose(dbeta, dy.sum(axis=(0, 1, 2)))
    assert np.allclose(dx.sum(axis=(0, 1, 2)), 0, atol=1e-10)
    # component along xhat: exactly eps/(var+eps) of the incoming one survives
    proj_in, proj_out = (dy * cache[0]).sum(axis=(0, 1, 2)), (dx * cache[0]).sum(axis=(0, 1, 2))
    assert np.allclose(proj_out, cache[1] * proj_in * (1e-3 / (var + 1e-3)))


def test_xavier_limits_for_weights_and_biases():
    """App. A-4: +-sqrt(6/(fan_in+fan_out)); a 1-D bias of n has fan_in=fan_out=n -> +-sqrt(3/n)."""
    rng = np.random.default_rng(3)
    w = T.xavier_uniform((5, 5, 3, 200), rng)
    assert np.abs(w).max() <= math.sqrt(6.0 / (75 + 5000))
    b = T.xavier_uniform((200,), rng)
    assert np.abs(b).max() <= math.sqrt(3.0 / 200) and np.abs(b).max() > 0.5 * math.sqrt(3.0 / 200)


def test_optimizer_first_step_closed_forms():
    """App. A-5: Adam's first step is lr*g/(|g| + eps*sqrt(1-b2))-ish; RMSProp rms slot starts at 1."""
    g = np.array([0.5, -2.0, 0.0])
    p = {'w': np.zeros(3)}
    T.Adam(1e-3, 0.5, 0.9).apply(p, {'w': g})
    lr_t = 1e-3 * math.sqrt(1 - 0.9) / (1 - 0.5)
    assert np.allclose(p['w'], -lr_t * (0.5 * g) / (np.sqrt(0.1 * g * g) + 1e-8))
    p = {'w': np.zeros(3)}
    T.RMSProp(1e-3, 0.9, 0.01).apply(p, {'w': g})
    assert np.allclose(p['w'], -1e-3 * g / np.sqrt(0.9 + 0.1 * g * g + 1e-10))
    p = {'w': np.ones(3)}
    o = T.Momentum(0.1, 0.5)
    o.apply(p, {'w': g})
    o.apply(p, {'w': g})
    assert np.allclose(p['w'], 1 - 0.1 * g - 0.1 * 1.5 * g)


def test_average_gradients_is_the_arithmetic_mean():
    """util.py:118-147."""
    t0, t1 = {'a': np.array([1.0, 3.0])}, {'a': np.array([3.0, 5.0])}
    assert np.allclose(T.average_gradients([t0, t1])['a'], [2.0, 4.0])


def test_sigmoid_xent_and_rescale():
    z, l = np.array([-3.0, 0.0, 2.0]), np.array([1.0, 0.0, 1.0])
    p = 1 / (1 + np.exp(-z))
    assert np.allclose(T.sigmoid_cross_entropy_with_logits(z, l), -(l * np.log(p) + (1 - l) * np.log(1 - p)))
    assert np.allclose(T.rescale(np.array([0.0, 0.5, 1.0]), (0, 1), (-1, 1)), [-1, 0, 1])     # hem/ops/images.py:68


def test_gradient_penalty_whole_batch_norm_quirk():
    """models/gan.py:229: ONE norm over the whole batch.  For a critic that is linear in its input
    (all pre-activations positive -> lrelu is the identity), grad = w_eff for every row, so
    slopes = sqrt(B) * ||w_eff||; a per-sample penalty would use ||w_eff||."""
    B, L = 4, 8
    cfg = G.make_cfg('iwgan', (32, 32, 3), L, B)
    P = G.init_params(cfg, 0, np.float64)
    for k in P:                                   # positive weights + large positive biases keep every unit active
        if 'discriminator' in k:
            P[k] = np.abs(P[k]) * 0.1 if k.endswith('weights') else np.full_like(P[k], 50.0)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (B, 3072))
    g = rng.uniform(-1, 1, (B, 3072))
    alpha = rng.uniform(0, 1, (B, 1))
    # w_eff from a single probe row
    _, cache = G.d_forward(P, x[:1], cfg)
    assert all((cache['pre%d' % i] > 0).all() for i in range(3))
    w_eff, _, _ = G.d_backward(P, cache, np.ones(1), cfg, want_params=False)
    gp, _ = G.gradient_penalty(P, x, g, alpha, cfg, want_param_grads=False)
    s = math.sqrt(B) * np.linalg.norm(w_eff)
    assert np.allclose(gp, (s - 1.0) ** 2)
    assert not np.allclose(gp, (np.linalg.norm(w_eff) - 1.0) ** 2)


def test_discriminator_row_count_follows_the_literal_reshape():
    """App. C-2: 64x64 input -> 8x8x4L features reshaped to [-1, 64L] = 4 rows per image."""
    assert G.make_cfg('iwgan', (32, 32, 3), 8, 2).d_rows_per_image == 1
    cfg = G.make_cfg('iwgan', (64, 64, 3), 8, 2)
    assert cfg.d_rows_per_image == 4
    P = G.init_params(cfg, 0, np.float64)
    d, _ = G.d_forward(P, np.zeros((2, 64 * 64 * 3)), cfg)
    assert d.shape == (8,)
