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


def test_two_replicas_match_one(tmp_path):
    worker = os.path.join(ROOT, 'tests', '_dist_worker.py')
    env = dict(os.environ, TDG_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    one, two = str(tmp_path / 'one.npz'), str(tmp_path / 'two.npz')
    env1 = {k: v for k, v in env.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    subprocess.run([sys.executable, worker, one], check=True, env=env1, timeout=600)
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                    '127.0.0.1', '--master-port', '29533', worker, two], check=True, env=env1, timeout=600)
    a, b = np.load(one), np.load(two)
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
