"""N > 1 path on CPU: two `gloo` ranks average a flat gradient bucket exactly like the reference's
tower mean (util.py:118-147) and start from the rank-0 variables."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from oracle import tf_ops as T


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    rt = importlib.import_module('3dgan_amd.runtime')
    eng = importlib.import_module('3dgan_amd.engine')
    util = importlib.import_module('3dgan_amd.util')
    assert rt.init_distributed('gloo') == world
    sess = rt.Session(device='cpu', dtype=0, seed=0)
    assert (sess.rank, sess.world_size) == (rank, world)
    store = eng.ParamStore('cpu')
    store.declare('discriminator/vars/c1/weights', (5, 5, 3, 8))
    store.declare('discriminator/vars/c1/bias', (8,))
    store.allocate()
    rng = np.random.default_rng(100 + rank)
    store.params.copy_(torch.tensor(rng.standard_normal(store.size), dtype=torch.float32))
    rt.broadcast_store(store)                                   # shared variables of the towers
    store.grads.copy_(torch.tensor(np.random.default_rng(rank).standard_normal(store.size), dtype=torch.float32))
    scale = util.average_gradients(sess, store)
    np.save(os.path.join(out_dir, 'r%d.npy' % rank),
            np.stack([store.params.numpy(), store.grads.numpy() * scale]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_mean_and_broadcast(tmp_path):
    world, port = 2, 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / 'r0.npy'), np.load(tmp_path / 'r1.npy')
    assert np.array_equal(r0[0], r1[0])                         # identical replicas after broadcast
    n = r0.shape[1]
    want = T.average_gradients([{'g': np.random.default_rng(0).standard_normal(n).astype(np.float32)},
                                {'g': np.random.default_rng(1).standard_normal(n).astype(np.float32)}])['g']
    assert np.allclose(r0[1], want, atol=1e-6) and np.array_equal(r0[1], r1[1])


def _split_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    rt = importlib.import_module('3dgan_amd.runtime')
    eng = importlib.import_module('3dgan_amd.engine')
    assert rt.init_distributed('gloo') == world
    sess = rt.Session(device='cpu', dtype=0, seed=0)
    # a critic-shaped bucket: [c1 | c2 | c3 weights + bias (the big slice, in the middle) | fc2]
    store = eng.ParamStore('cpu')
    for name, shape in (('c1/weights', (5, 5, 3, 8)), ('c1/bias', (8,)), ('c2/weights', (5, 5, 8, 16)), ('c2/bias', (16,)),
                        ('c3/weights', (5, 5, 16, 30)), ('c3/bias', (30,)), ('fc2/weights', (480, 1)), ('fc2/bias', (1,))):
        store.declare('discriminator/vars/' + name, shape)
    store.allocate()
    lo = store.index['discriminator/vars/c3/weights'][0]
    hi = store.index['discriminator/vars/c3/bias'][0] + (30 + 3) // 4 * 4
    assert 0 < lo < hi < store.size
    g = torch.tensor(np.random.default_rng(50 + rank).standard_normal(store.size), dtype=torch.float32)
    # (1) the split exchange of GanReplica.d_step; the rest of the bucket is only FINISHED inside `between`
    store.grads.copy_(g)
    store.grads[:lo] = 0
    store.grads[hi:] = 0
    ran = []

    def between():
        ran.append(1)
        store.grads[:lo] = g[:lo]
        store.grads[hi:] = g[hi:]
    scale = sess.allreduce_split(store.grads, lo, hi, between=between)
    split = store.grads.numpy().copy() * scale
    # (2) one all-reduce of the whole bucket
    store.grads.copy_(g)
    scale1 = sess.allreduce_mean_scale(store.grads)
    whole = store.grads.numpy().copy() * scale1
    # (3) a NaN on ONE replica raises on EVERY replica once the slices are summed (--check_numerics)
    sess.check_numerics = True
    # (the flag itself is a HIP kernel, tdg_check_finite, covered by tests/test_gpu_parity_holes.py; here a host stand-in)
    sess._nonfinite_flag = lambda st: torch.tensor([0 if bool(torch.isfinite(st.grads).all()) else 1], dtype=torch.int32)
    store.grads.copy_(g)
    if rank == 1:
        store.grad('discriminator/vars/c2/weights').view(-1)[3] = float('nan')
    sess.allreduce_split(store.grads, lo, hi)
    try:
        sess.assert_finite(store, 'd_step')
        raised = ''
    except FloatingPointError as e:
        raised = str(e)
    # (4) the loss scalars every rank reports are the LAST replica's (util.py:187-193); --mean_loss averages
    scal = torch.full((16,), float(rank + 1))
    last = sess.report_scalars(scal).numpy().copy()
    mean = sess.report_scalars(scal, mean=True).numpy().copy()
    np.savez(os.path.join(out_dir, 'r%d.npz' % rank), split=split, whole=whole, ran=np.array(len(ran)), raised=np.array(raised),
             last=last, mean=mean, scale=np.array([scale, scale1]))
    dist.barrier()
    dist.destroy_process_group()


def test_split_bucket_exchange_equals_one_allreduce(tmp_path):
    """GanReplica.d_step's exchange (big slice first and asynchronous, the rest after the work that completes it) is the
    same mean over towers as one all-reduce of the flat bucket (util.py:118-147), on both replicas."""
    world, port = 2, 31000 + os.getpid() % 2000
    mp.spawn(_split_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ('r%d.npz' % i)) for i in range(world)]
    for z in r:
        assert int(z['ran']) == 1 and list(z['scale']) == [0.5, 0.5]
        assert np.array_equal(z['split'], z['whole'])
        assert 'd_step' in str(z['raised']) and 'c2/weights' in str(z['raised'])      # both replicas name the variable
        assert np.all(z['last'] == 2.0) and np.all(z['mean'] == 1.5)
    assert np.array_equal(r[0]['split'], r[1]['split'])
    n = r[0]['split'].size
    want = 0.5 * (np.random.default_rng(50).standard_normal(n).astype(np.float32) +
                  np.random.default_rng(51).standard_normal(n).astype(np.float32))
    assert np.allclose(r[0]['split'], want, atol=1e-6)
