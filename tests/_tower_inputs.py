"""Inputs of the multi-replica rehearsals (tests/test_gpu_distributed.py, tests/_dist_worker.py): every replica
("tower", util.py:54-77 of the reference) gets its OWN batch shard (ops/input.py:11-25) and its OWN z / alpha / eps
draws (models/gan.py:246,224 and models/vae.py:127 sit inside the tower loop).  The worker processes and the test
process (which feeds the oracle) derive them from the same seeds here."""
from types import SimpleNamespace

import numpy as np

SIZES = {'iwgan': dict(B=8, L=16, shape=(32, 32, 3)), 'wgan': dict(B=8, L=16, shape=(32, 32, 3)),
         'vae': dict(B=8, L=16, shape=(64, 64, 3)), 'pix2pix': dict(B=1, L=0, shape=(256, 256, 3))}
ITERATIONS = 3            # eager warm-up, hipGraph capture, replay
N_DISC = 2


def make_args(model, world, use_graphs=True):
    s = SIZES[model]
    if model == 'vae':
        return SimpleNamespace(model='vae', batch_size=s['B'], latent_size=s['L'], image_shape=s['shape'], n_gpus=world,
                               optimizer='rmsprop', lr=1e-3, beta1=0.9, beta2=0.999, decay=0.9, momentum=0.01, centered=False,
                               use_graphs=use_graphs)              # config 5: train.py's optimizer defaults (train.py:113-136)
    if model == 'pix2pix':
        return SimpleNamespace(model='pix2pix', batch_size=s['B'], n_gpus=world, optimizer='adam', lr=1e-4, beta1=0.5,
                               beta2=0.999, decay=0.9, momentum=0.01, centered=False, n_disc_train=1, skip_layers=True,
                               noise=[], dropout=0, batch_norm_disc=False, batch_norm_gen=False, add_l1=True, seed=3,
                               use_graphs=use_graphs)
    opt = (dict(optimizer='adam', lr=1e-4, beta1=0.5, beta2=0.9) if model == 'iwgan' else
           dict(optimizer='rmsprop', lr=5e-5, beta1=0.9, beta2=0.999))
    return SimpleNamespace(model=model, batch_size=s['B'], latent_size=s['L'], image_shape=s['shape'], n_gpus=world,
                           decay=0.9, momentum=0.01, centered=False, n_disc_train=N_DISC, display_d_loss=True,
                           use_graphs=use_graphs, **opt)


def iterations(model):
    """Staged (oracle) runs: pix2pix stops after the capture iteration (its replays are covered bit for bit by the
    shards-vs-towers rehearsal; every further iteration is 220 MB of generator gradients to carry)."""
    return 2 if model == 'pix2pix' else ITERATIONS


def steps_per_iteration(model):
    return {'iwgan': N_DISC + 1, 'wgan': N_DISC + 1, 'vae': 1, 'pix2pix': 3}[model]


def step_inputs(model, rank, step):
    """What tower `rank` consumes in optimizer step (or report pass) number `step` of the run: a dict with 'x' (and
    'y' for pix2pix) in [0, 1] and the draws 'z', 'alpha' / 'eps' the model makes, all float32."""
    s = SIZES[model]
    rng = np.random.default_rng([1234, rank, step])
    B, L, (h, w, c) = s['B'], s['L'], s['shape']
    out = {'x': rng.uniform(0, 1, (B, h, w, c)).astype(np.float32)}
    if model == 'pix2pix':
        out['y'] = (rng.uniform(0, 1, (B, h, w, 1)) * 0.98 + 0.01).astype(np.float32)
    elif model == 'vae':
        out['eps'] = rng.standard_normal((B, L)).astype(np.float32)
    else:
        out['z'] = rng.standard_normal((B, L)).astype(np.float32)
        out['alpha'] = rng.uniform(0, 1, (B, 1)).astype(np.float32)
    return out
