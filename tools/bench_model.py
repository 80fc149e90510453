"""Diagnostics: step time and per-kernel GEMM breakdown of the secondary models (vae | cnn | pix2pix) on synthetic data.
usage: python tools/bench_model.py MODEL [batch] [steps]"""
import importlib
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module('3dgan_amd.kernels')
rt = importlib.import_module('3dgan_amd.runtime')
data = importlib.import_module('3dgan_amd.data')
models = importlib.import_module('3dgan_amd.models')


def main():
    model = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if model == 'pix2pix' else 512)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    sess = rt.Session(dtype=K.BF16, seed=0, rank=0, world_size=1)
    shape = (256, 256, 3) if model == 'pix2pix' else (64, 64, 3)
    args = SimpleNamespace(model=model, batch_size=B, latent_size=200, image_shape=shape, n_gpus=1, optimizer='adam', lr=1e-4,
                           beta1=0.5, beta2=0.999, decay=0.9, momentum=0.01, centered=False, n_disc_train=1, skip_layers=True,
                           noise=[], dropout=0, batch_norm_disc=False, batch_norm_gen=False, add_l1=False, seed=0)
    src = data.SyntheticPairSource(2, B, sess.device) if model == 'pix2pix' else data.SyntheticSource(2 * B, shape, B, sess.device)
    train = models.model_funcs()[model](src, args, sess)
    for _ in range(3):
        train(sess, args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        train(sess, args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('%s batch %d: %.2f ms per train() call, %.0f images/s' % (model, B, dt * 1e3, B / dt))
    rep = train.replica
    if hasattr(rep, 'use_graphs'):
        rep.use_graphs = False
    train(sess, args)
    K.timing_begin()
    train(sess, args)
    torch.cuda.synchronize()
    rec = K.timing_end()
    acc = {}
    for name, ms, fl in rec:
        e = acc.setdefault(name, [0, 0.0, 0.0])
        e[0] += 1; e[1] += ms; e[2] += fl
    tot = sum(v[1] for v in acc.values())
    print('conv GEMM kernels: %.2f ms per call' % tot)
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print('  %-44s %4d launches %8.3f ms %7.1f TF' % (k, v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12 if v[1] else 0))
    if os.environ.get('TDG_LIST_LAUNCHES'):
        for i, (name, ms, fl) in enumerate(rec):
            print('  #%03d %-44s %8.4f ms %8.2f GFLOP %7.1f TF' % (i, name, ms, fl / 1e9, fl / (ms * 1e-3) / 1e12 if ms else 0))


if __name__ == '__main__':
    main()
