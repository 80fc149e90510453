"""Diagnostics: what the N > 1 schedule of the headline step costs on ONE GPU.  A one-rank `nccl` (RCCL) process group makes every
all-reduce the identity, while the replica is told it is one of `world` replicas and therefore takes the split critic bodies, the
asynchronous slice exchange and the overlapped generator exchange.  Compare with the same build at world = 1.
usage: python tools/bench_exchange_path.py [world=8] [steps=10]"""
import importlib
import os
import sys
import time
from types import SimpleNamespace

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module('3dgan_amd.kernels')
rt = importlib.import_module('3dgan_amd.runtime')
gan = importlib.import_module('3dgan_amd.models.gan')
data = importlib.import_module('3dgan_amd.data')


def run(world, steps):
    sess = rt.Session(device=torch.device('cuda:0'), dtype=K.BF16, seed=0, rank=0, world_size=world)
    args = SimpleNamespace(model='iwgan', batch_size=512, latent_size=200, image_shape=(32, 32, 3), n_gpus=world, optimizer='adam', lr=1e-4,
                           beta1=0.5, beta2=0.9, decay=0.9, momentum=0.01, centered=False, n_disc_train=5, display_d_loss=True, use_graphs=True)
    rep = gan.GanReplica(data.SyntheticSource(12 * 512, (32, 32, 3), 512, sess.device, 1234, 0), args, sess)
    for _ in range(4):
        rep.train_func()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rep.train_func()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if __name__ == '__main__':
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29561')
    torch.cuda.set_device(0)
    dist.init_process_group(backend=os.environ.get('TDG_DIST_BACKEND', 'nccl'), rank=0, world_size=1)
    one = run(1, steps)
    many = run(world, steps)
    print('headline step on one GPU: %.2f ms as a single replica, %.2f ms on the %d-replica schedule over a one-rank %s group '
          '(identity all-reduces): the schedule itself costs %+.2f ms per iteration' % (one, many, world, dist.get_backend(), many - one))
    dist.destroy_process_group()
