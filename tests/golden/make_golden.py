"""Generates the golden vectors under tests/golden/ from the CPU oracle (float64 arithmetic,
stored as float32 inputs / float64 expectations).  The reference cannot be imported here
(TensorFlow is not installed: ModuleNotFoundError) and holds no vectors for this path, so
these fixtures pin the *oracle* (and through it the HIP path), not TensorFlow: parity unpinned.

    python tests/golden/make_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gan_ref as G  # noqa: E402
from oracle import tf_ops as T  # noqa: E402


def gan_case(model, seed):
    B, L, shape = 4, 8, (32, 32, 3)
    cfg = G.make_cfg(model, shape, L, B)
    P32 = G.init_params(cfg, seed, np.float32)
    rng = np.random.default_rng(seed + 100)
    xs = [rng.uniform(0, 1, (B,) + shape).astype(np.float32) for _ in range(2)]
    zs = [rng.standard_normal((B, L)).astype(np.float32) for _ in range(2)]
    als = [rng.uniform(0, 1, (B, 1)).astype(np.float32) for _ in range(2)]
    args = SimpleNamespace(optimizer='rmsprop', lr=1e-3, decay=0.9, momentum=0.01, centered=False, beta1=0.5, beta2=0.9,
                           n_disc_train=1)
    P = {k: v.astype(np.float64) for k, v in P32.items()}
    tr = G.GanTrainer(P, cfg, args)
    x0 = tr.rescale(xs[0].astype(np.float64))
    d_loss0, d_grads, aux = G.d_loss_and_grads(P, x0, zs[0].astype(np.float64), als[0].astype(np.float64), cfg)
    out = tr.train_func([x.astype(np.float64) for x in xs], [z.astype(np.float64) for z in zs],
                        [a.astype(np.float64) for a in als])
    save = {'param/' + k: v for k, v in P32.items()}
    save.update({'x%d' % i: x for i, x in enumerate(xs)})
    save.update({'z%d' % i: z for i, z in enumerate(zs)})
    save.update({'alpha%d' % i: a for i, a in enumerate(als)})
    save.update({'dgrad/' + k: v.astype(np.float32) for k, v in d_grads.items()})
    save.update({'after/' + k: v.astype(np.float32) for k, v in tr.P.items()})
    save['d_loss0'] = np.array(d_loss0)
    save['gp0'] = np.array(aux['gp'])
    save['g_loss'] = np.array(out['g_loss'])
    save['d_loss'] = np.array(out['d_loss'])
    np.savez_compressed(os.path.join(HERE, '%s_step_L8_B4.npz' % model), **save)


def conv_case():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 8, 8, 8)).astype(np.float32)
    K = (rng.standard_normal((5, 5, 8, 16)) / 10).astype(np.float32)
    dy = rng.standard_normal((2, 4, 4, 16)).astype(np.float32)
    x64, K64, dy64 = x.astype(np.float64), K.astype(np.float64), dy.astype(np.float64)
    np.savez_compressed(os.path.join(HERE, 'conv_k5s2.npz'), x=x, K=K, dy=dy,
                        y=T.conv2d(x64, K64, 2),
                        dx=T.conv2d_backprop_input(x.shape, K64, dy64, 2),
                        dK=T.conv2d_backprop_filter(x64, K.shape, dy64, 2),
                        yt=T.conv2d_transpose(dy64, K64, (2, 8, 8, 8), 2))


if __name__ == '__main__':
    gan_case('iwgan', 0)
    gan_case('wgan', 1)
    conv_case()
    print('golden vectors written to', HERE)
