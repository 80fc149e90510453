"""The N > 1 path on the one-GPU box: two replica processes (torch.distributed.run, gloo standing in for RCCL, both on
cuda:0) stepping the iwgan schedule with the split D-gradient bodies and the early all-reduce of the largest filter's
slice.  With identical data and RNG keys in both replicas the tower mean equals each tower's gradient, so losses and
every variable must equal the single-replica run bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('model,port', [('iwgan', 29533), ('wgan', 29534), ('vae', 29535), ('pix2pix', 29536)])
def test_two_replicas_match_one(tmp_path, model, port):
    """iwgan: the headline schedule; wgan: config 3's model (rmsprop, one exchange per step); vae: config 5's model; pix2pix: config 4's
    (256 x 256, one pair per replica)."""
    worker = os.path.join(ROOT, 'tests', '_dist_worker.py')
    env = dict(os.environ, TDG_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    one, two = str(tmp_path / 'one.npz'), str(tmp_path / 'two.npz')
    env1 = {k: v for k, v in env.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    subprocess.run([sys.executable, worker, one, model], check=True, env=env1, timeout=600)
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                    '127.0.0.1', '--master-port', str(port), worker, two, model], check=True, env=env1, timeout=600)
    a, b = np.load(one), np.load(two)
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher environment starts its own two ranks (a child torch.distributed.run)
    and prints ONE JSON line carrying the ranks seen on the process group and the per-bucket all-reduce times."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(TDG_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--batch_size', '16', '--latent_size', '16', '--timer_steps', '1'],
                       env=env, timeout=900, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['ranks_seen'] == 2 and out['scaling'] == 'weak'
    assert out['config']['global_batch'] == 32
    c = out['collectives']
    assert c['ranks_seen'] == 2 and c['backend'] == 'gloo'
    for k in ('d_big_slice', 'd_rest', 'g_bucket'):
        assert c[k]['ms'] > 0 and c[k]['bytes'] > 0
    assert 'roofline' in out and 'cpu_baseline' not in out


@pytest.mark.parametrize('model', ['iwgan', 'vae'])
def test_rccl_runs_the_exchange_path(tmp_path, model):
    """RCCL itself on the one-GPU box: a ONE-rank `nccl` process group (the only RCCL configuration one device allows) under a
    replica that takes its N > 1 code path (split critic bodies, asynchronous slice all-reduce, 1/n in the optimizer, hipGraph
    capture beside RCCL's watchdog thread).  With one rank every all-reduce is the identity, so the run must equal the same
    run over a one-rank gloo group bit for bit."""
    worker = os.path.join(ROOT, 'tests', '_dist_worker.py')
    base = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    outs = {}
    for i, backend in enumerate(('gloo', 'nccl')):
        outs[backend] = str(tmp_path / (backend + '.npz'))
        env = dict(base, TDG_DIST_BACKEND=backend, TDG_FAKE_WORLD='2', TDG_PORT=str(29551 + i), HSA_ENABLE_IPC_MODE_LEGACY='0')
        p = subprocess.run([sys.executable, worker, outs[backend], model], env=env, timeout=600, capture_output=True, text=True)
        assert p.returncode == 0, (backend, p.stderr[-3000:])
    a, b = np.load(outs['gloo']), np.load(outs['nccl'])
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.all(np.isfinite(a[k])), k
        assert np.array_equal(a[k], b[k]), k


def test_train_cli_launches_its_replicas(tmp_path):
    """`python train.py --n_gpus 2` (the reference's flag, train.py:262) starts one process per replica itself, trains an epoch and
    leaves rank 0's checkpoint + options.config behind (gloo standing in for RCCL on the one-GPU box)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(TDG_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    d = str(tmp_path / 'ws')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '--model', 'iwgan', '--batch_size', '8', '--latent_size', '16',
                        '--optimizer', 'adam', '--lr', '1e-4', '--beta1', '0.5', '--beta2', '0.9', '--dataset', 'synthetic',
                        '--epoch_size', '3', '--epochs', '1', '--n_gpus', '2', '--dir', d],
                       env=env, timeout=900, capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert 'Training complete' in p.stdout
    found = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs]
    assert any(f.endswith('options.config') for f in found) and any('checkpoint' in os.path.basename(f) for f in found), found
