"""GPU parity of a gen-2 VALID-padded conv / deconv stack built through the layer builders.

The geometry is hem/models/paper_cgan.py:212-243 `g_baseline` without its skip concatenations (a chain):
encoder conv k5 s2 VALID relu 65 -> 31 -> 14 -> 5 -> 1, decoder deconv k5 s2 VALID with explicit output_shape
1 -> 5 -> 14 -> 31 (lrelu 0.2), then a 1x1 SAME conv without activation.  Forward output and every gradient of a
linear functional of the output are compared with torch autograd in float64 (f32 path: max-norm 1e-3; bf16: l2 0.3)."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T
from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu

ENC = [('e1', 3, 16), ('e2', 16, 24), ('e3', 24, 40), ('e4', 40, 64)]
DEC = [('d1', 64, 40, 5), ('d2', 40, 24, 14), ('d3', 24, 16, 31)]


def build(B):
    Lm, act = pkg('ops.layers'), pkg('ops.activations')
    Lm.reset_graph()
    x = Lm.placeholder((None, 65, 65, 3))
    with Lm.variable_scope('generator') as net:
        with Lm.arg_scope([Lm.conv2d], filter_size=5, stride=2, padding='VALID', activation=act.relu):
            h = x
            for name, ci, co in ENC:
                h = Lm.conv2d(h, ci, co, name=name)
        assert h.shape[1:] == (1, 1, 64)
        with Lm.arg_scope([Lm.deconv2d, Lm.conv2d], filter_size=5, stride=2, padding='VALID',
                          activation=lambda t: act.lrelu(t, leak=0.2)):
            for name, ci, co, size in DEC:
                h = Lm.deconv2d(h, ci, co, output_shape=(B, co, size, size), name=name)
            h = Lm.conv2d(h, 16, 1, stride=1, filter_size=1, padding='SAME', activation=None, name='d4')
        assert h.shape[1:] == (31, 31, 1)
    return net


def torch_forward(P, x):
    h = x
    for name, _, _ in ENC:
        h = torch.relu(TR.conv2d_valid(h, P['generator/vars/%s/weights' % name], 2) + P['generator/vars/%s/bias' % name])
    for name, _, _, size in DEC:
        h = TR.lrelu(TR.conv2d_transpose_valid(h, P['generator/vars/%s/weights' % name], (size, size))
                     + P['generator/vars/%s/bias' % name])
    return TR.conv2d_same(h, P['generator/vars/d4/weights'], 1) + P['generator/vars/d4/bias']


def relerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def l2err(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize('dtype', [0, 1])
def test_valid_stack_forward_and_gradients(dtype):
    K, E = pkg('kernels'), pkg('engine')
    dev = torch.device('cuda:0')
    B = 3
    net = build(B)
    store = E.ParamStore(dev)
    seq = E.SeqNet(net, B, (65, 65, 3), dtype, dev, store, need_input_grad=True)
    seq.declare_variables()
    store.allocate()
    rng = np.random.default_rng(4)
    P0 = {k: T.xavier_uniform(shape, rng, np.float64) for k, (_, shape) in store.index.items()}
    assert set(P0) == {'generator/vars/%s/%s' % (n, v) for n in ('e1', 'e2', 'e3', 'e4', 'd1', 'd2', 'd3', 'd4')
                       for v in ('weights', 'bias')}
    assert store.index['generator/vars/d2/weights'][1] == (5, 5, 24, 40)          # [k, k, Cout, Cin]
    store.load(P0)
    seq.repack()
    x = rng.uniform(-1, 1, (B, 65, 65, 3))
    R = rng.standard_normal((B, 31, 31, 1))
    seq.x.set(x.astype(np.float32))
    out = seq.forward(0, B)
    last = seq.layers[-1]
    last.gout.set(R.astype(np.float32))
    seq.backward(0, B, want_params=True, want_dx=True)

    Pt = {k: torch.tensor(v, requires_grad=True) for k, v in P0.items()}
    xt = torch.tensor(x, requires_grad=True)
    yt = torch_forward(Pt, xt)
    (yt * torch.tensor(R)).sum().backward()
    # f32: max-norm 1e-3.  bf16: the 1x1x64 bottleneck makes single relu flips visible in the deepest gradients, so
    # the bf16 run is only held to a loose l2 bound (it exercises the chain in bf16; per-kernel bf16 parity on these
    # geometries is tests/test_gpu_kernels.py::test_conv_valid_padding_all_forms)
    err, tol = (relerr, 1e-3) if dtype == 0 else (l2err, 0.3)
    assert err(out.get(), yt.detach().numpy()) < tol
    got = store.grads_dict()
    for k, p in Pt.items():
        assert err(got[k], p.grad.numpy()) < tol, k
    assert err(seq.dx.get(), xt.grad.numpy()) < tol
