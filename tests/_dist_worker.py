"""Worker of tests/test_gpu_distributed.py: one replica of a 2-replica run (launched by torch.distributed.run) of iwgan, wgan, vae or pix2pix.
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


def run(out_path, world, model='iwgan'):
    K = importlib.import_module('3dgan_amd.kernels')
    rt = importlib.import_module('3dgan_amd.runtime')
    data = importlib.import_module('3dgan_amd.data')
    fake = int(os.environ.get('TDG_FAKE_WORLD', '0'))       # one-rank process group, but the replica takes its N > 1 code path
    if fake:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', os.environ.get('TDG_PORT', '29541'))
        torch.cuda.set_device(rt.local_device())
        dist.init_process_group(backend=os.environ['TDG_DIST_BACKEND'], rank=0, world_size=1)
        world = fake
    elif world > 1:
        rt.init_distributed()
    sess = rt.Session(device=rt.local_device(), dtype=K.BF16, seed=3, rank=0, world_size=world)
    if model == 'vae':                                        # config 5's model (BASELINE.json configs[4]); same exchange, one bucket
        vae = importlib.import_module('3dgan_amd.models.vae')
        B, L, shape = 8, 16, (64, 64, 3)
        args = SimpleNamespace(model='vae', batch_size=B, latent_size=L, image_shape=shape, n_gpus=world, optimizer='adam',
                               lr=1e-3, beta1=0.9, beta2=0.999, decay=0.9, momentum=0.01, centered=False, use_graphs=True)
        rep = vae.VaeReplica(data.SyntheticSource(6 * B, shape, B, sess.device, 5, 0), args, sess)
        stores = [rep.store]
    elif model == 'pix2pix':                                  # config 4's model: two nets, one exchange per optimizer step
        p2p = importlib.import_module('3dgan_amd.models.pix2pix')
        B = 1
        args = SimpleNamespace(model='pix2pix', batch_size=B, n_gpus=world, optimizer='adam', lr=1e-4, beta1=0.5, beta2=0.999, decay=0.9,
                               momentum=0.01, centered=False, n_disc_train=1, skip_layers=True, noise=[], dropout=0, batch_norm_disc=False,
                               batch_norm_gen=False, add_l1=True, seed=3, use_graphs=True)
        rep = p2p.pix2pix(data.SyntheticPairSource(3, B, sess.device, seed=5, rank=0), args, sess)
        rep.train_func = lambda: rep.train(sess, args, None)
        stores = rep.stores()
    else:                                                     # iwgan (split critic exchange) / wgan (config 3's model: rmsprop, one exchange)
        gan = importlib.import_module('3dgan_amd.models.gan')
        B, L, shape = 8, 16, (32, 32, 3)
        opt = dict(optimizer='adam', lr=1e-4, beta1=0.5, beta2=0.9) if model == 'iwgan' else dict(optimizer='rmsprop', lr=5e-5, beta1=0.9, beta2=0.999)
        args = SimpleNamespace(model=model, batch_size=B, latent_size=L, image_shape=shape, n_gpus=world, decay=0.9, momentum=0.01,
                               centered=False, n_disc_train=2, display_d_loss=True, use_graphs=True, **opt)
        rep = gan.GanReplica(data.SyntheticSource(6 * B, shape, B, sess.device, 5, 0), args, sess)
        stores = rep.stores()
    for s in stores:
        rt.broadcast_store(s)
    rep.refresh()
    losses = [rep.train_func() for _ in range(4)]          # eager, capture, 2 replays
    torch.cuda.synchronize()
    if int(os.environ.get('RANK', '0')) == 0:
        np.savez(out_path, **{'loss_%s' % k: np.array([l[k] for l in losses]) for k in sorted(losses[0])},
                 **{k.replace('/', '.'): v for k, v in rep.variables().items()})
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    run(sys.argv[1], int(os.environ.get('WORLD_SIZE', '1')), sys.argv[2] if len(sys.argv) > 2 else 'iwgan')
