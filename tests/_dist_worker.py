"""Worker of tests/test_gpu_distributed.py: one replica of a 2-replica run (launched by torch.distributed.run).
Both replicas are given the SAME data, seed and RNG key (Session(rank=0)), so the mean over replicas equals each
replica's own gradient exactly and the variables must match a single-replica run bit for bit."""
import importlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(out_path, world):
    K = importlib.import_module('3dgan_amd.kernels')
    rt = importlib.import_module('3dgan_amd.runtime')
    gan = importlib.import_module('3dgan_amd.models.gan')
    data = importlib.import_module('3dgan_amd.data')
    if world > 1:
        rt.init_distributed()
    B, L, shape = 8, 16, (32, 32, 3)
    args = SimpleNamespace(model='iwgan', batch_size=B, latent_size=L, image_shape=shape, n_gpus=world, optimizer='adam',
                           lr=1e-4, beta1=0.5, beta2=0.9, decay=0.9, momentum=0.01, centered=False, n_disc_train=2,
                           display_d_loss=True, use_graphs=True)
    sess = rt.Session(device=rt.local_device(), dtype=K.BF16, seed=3, rank=0, world_size=world)
    rep = gan.GanReplica(data.SyntheticSource(6 * B, shape, B, sess.device, 5, 0), args, sess)
    for s in rep.stores():
        rt.broadcast_store(s)
    rep.refresh()
    losses = [rep.train_func() for _ in range(4)]          # eager, capture, 2 replays
    torch.cuda.synchronize()
    if int(os.environ.get('RANK', '0')) == 0:
        np.savez(out_path, g_loss=np.array([l['g_loss'] for l in losses]), d_loss=np.array([l['d_loss'] for l in losses]),
                 **{k.replace('/', '.'): v for k, v in rep.variables().items()})
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    run(sys.argv[1], int(os.environ.get('WORLD_SIZE', '1')))
