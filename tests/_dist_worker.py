"""Worker of tests/test_gpu_distributed.py: one process of a multi-replica rehearsal of iwgan, wgan, vae or pix2pix
(launched by torch.distributed.run for the two-replica runs, directly for the single-process ones).

Modes (third argument):
  same    both replicas are given the SAME data, seed and RNG key (Session(rank=0)): the mean over replicas equals each
          replica's own gradient exactly, so the variables must match a single-replica run bit for bit (a wrong 1/n shows).
  shards  every replica has its OWN batch shard and RNG key (the reference's towers: ops/input.py:11-25,
          models/gan.py:246,224 inside the tower loop): bf16, hipGraphs, on-device Philox draws.
  towers  ONE process, no process group: the same two towers run one after the other on the device, their gradient
          buckets are added and both apply the mean -- "n independent replicas, then mean"  with the HIP path
          itself as the replica.  a + b is commutative in floating point, so `shards` must equal this bit for bit.
  staged  as `shards`, but f32 with the z / alpha / eps draws of tests/_tower_inputs.py staged on the device, so the
          test process can run the float64 oracle (oracle/towers_ref.py) on identical inputs.

TDG_TEST_SABOTAGE (negative tests: the comparison must turn red): `rest` drops the exchange of the critic bucket's
remainder around the early slice, `g_async` drops the generator bucket's asynchronous exchange, `bucket` drops every
whole-bucket exchange."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import _tower_inputs as TI                                   # noqa: E402


def _sabotage(rt, kind):
    import torch.distributed as dist
    S = rt.Session
    if kind == 'rest':
        def split(self, flat_grads, lo, hi, between=None):
            work = self.allreduce_async(flat_grads[lo:hi])
            if between is not None:
                between()
            if work is not None:
                work.wait()
            return 1.0 / self.world_size                      # [:lo] and [hi:] are never summed
        S.allreduce_split = split
    elif kind == 'g_async':
        orig_split, orig_async = S.allreduce_split, S.allreduce_async
        state = {'in_split': False}

        def split(self, *a, **k):
            state['in_split'] = True
            try:
                return orig_split(self, *a, **k)
            finally:
                state['in_split'] = False

        class Done:
            def wait(self):
                return True

        def async_(self, flat_slice):
            if state['in_split']:
                return orig_async(self, flat_slice)
            return Done() if self.world_size > 1 else None   # the generator bucket stays local
        S.allreduce_split, S.allreduce_async = split, async_
    elif kind == 'bucket':
        S.allreduce_mean_scale = lambda self, flat: (1.0 / self.world_size if self.world_size > 1 else 1.0)
    elif kind:
        raise ValueError(kind)


def _build(model, world, sess, source, use_graphs=True):
    """(replica, stores) of one tower."""
    args = TI.make_args(model, world, use_graphs)
    if model == 'vae':
        rep = importlib.import_module('3dgan_amd.models.vae').VaeReplica(source, args, sess)
        return rep, [rep.store]
    if model == 'pix2pix':
        rep = importlib.import_module('3dgan_amd.models.pix2pix').pix2pix(source, args, sess)
        rep.train_func = lambda: rep.train(sess, args, None)
        return rep, rep.stores()
    rep = importlib.import_module('3dgan_amd.models.gan').GanReplica(source, args, sess)
    return rep, rep.stores()


def _source(model, sess, data_rank):
    data = importlib.import_module('3dgan_amd.data')
    s = TI.SIZES[model]
    if model == 'pix2pix':
        return data.SyntheticPairSource(3, s['B'], sess.device, seed=5, rank=data_rank)
    return data.SyntheticSource(6 * s['B'], s['shape'], s['B'], sess.device, 5, data_rank)


def _phases(model, rep):
    """The optimizer steps of one training iteration as (compute gradients of a batch, bucket, apply) triples + the
    report, through the replica's own captured bodies (what d_step / g_step / step run around their exchange)."""
    if model == 'vae':
        def grads(b):
            rep.x_stage.copy_(b.reshape(rep.x_stage.shape))
            rep._run('grads', rep._grads)
        return [(grads, rep.store, lambda: rep._run('apply', rep._apply))], (lambda b: rep.losses())
    if model == 'pix2pix':
        def dg(b):
            rep._stage(b)
            rep._run('d_grads', rep._d_grads)

        def gg(b):
            rep._stage(b)
            rep._run('g_grads', rep._g_grads)
        return ([(dg, rep.d_store, lambda: rep._run('d_apply', rep._d_apply)),
                 (gg, rep.g_store, lambda: rep._run('g_apply', rep._g_apply))], rep.report)

    nd = rep.args.n_disc_train

    def dgr(i):
        def grads(b):                                                 # what train_func + d_step do around the critic's gradient body
            rep._load_real(b)
            if i == 0:
                rep._iter_ahead = rep._begin_ahead(nd)                # the iteration's generator passes, taken ahead
            rep._ahead = i if rep._iter_ahead else None
            try:
                rep._run('d_grads' + rep._stage_ahead(), rep._d_grads)
            finally:
                rep._ahead = None
        return grads

    def ggr(b):
        rep._load_real(b)
        rep._run('g_grads', rep._g_grads)
    d = [(dgr(i), rep.d_store, lambda: rep._run('d_apply', rep._d_apply)) for i in range(nd)]
    g = (ggr, rep.g_store, lambda: rep._run('g_apply', rep._g_apply))
    return d + [g], (lambda b: rep.losses())


def _save(out_path, rep, losses, extra=None):
    np.savez(out_path, **{'loss_%s' % k: np.array([l[k] for l in losses]) for k in sorted(losses[0])},
             **{k.replace('/', '.'): v for k, v in rep.variables().items()}, **(extra or {}))


def run_towers(out_path, model, n_towers=2):
    K = importlib.import_module('3dgan_amd.kernels')
    rt = importlib.import_module('3dgan_amd.runtime')
    dev = rt.local_device()
    reps = []
    for r in range(n_towers):
        sess = rt.Session(device=dev, dtype=K.BF16, seed=3, rank=r, world_size=1)
        rep, stores = _build(model, n_towers, sess, _source(model, sess, r))
        if reps:                                                       # shared variables: tower 0's initial values
            for s0, s in zip(reps[0][1], stores):
                s.params.copy_(s0.params)
            rep.refresh()
        reps.append((rep, stores))
    plans = [_phases(model, rep) for rep, _ in reps]
    losses = []
    for _ in range(TI.ITERATIONS + 1):
        for p in range(len(plans[0][0])):
            for (rep, _), (phases, _r) in zip(reps, plans):
                grads, store, _a = phases[p]
                grads(rep.x_y.next_batch() if model == 'pix2pix' else rep.x_source.next_batch())
            total = sum(pl[0][p][1].grads for pl in plans)          # what the all-reduce(sum) leaves in every bucket
            for (rep, _), (phases, _r) in zip(reps, plans):
                phases[p][1].grads.copy_(total)
                rep._scale = 1.0 / n_towers                          # average_gradients (util.py:138-139), in the optimizer kernel
                phases[p][2]()
        out = None
        for (rep, _), (_p, report) in zip(reps, plans):               # every tower evaluates its losses; the dict keeps the last
            out = report(rep.x_y.next_batch() if model == 'pix2pix' else None)
        losses.append(out)
    torch.cuda.synchronize()
    _save(out_path, reps[0][0], losses)


def run(out_path, world, model='iwgan', mode='same'):
    K = importlib.import_module('3dgan_amd.kernels')
    rt = importlib.import_module('3dgan_amd.runtime')
    _sabotage(rt, os.environ.get('TDG_TEST_SABOTAGE', ''))
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
    rank = int(os.environ.get('RANK', '0'))
    key_rank = 0 if mode == 'same' else rank                 # RNG key and data shard of this replica
    dtype = K.F32 if mode == 'staged' else K.BF16
    sess = rt.Session(device=rt.local_device(), dtype=dtype, seed=3, rank=key_rank, world_size=world)
    rep, stores = _build(model, world, sess, _source(model, sess, key_rank))
    for s in stores:
        rt.broadcast_store(s)
    rep.refresh()
    extra = {}
    if mode == 'staged':
        extra.update({'init.' + k.replace('/', '.'): v for k, v in rep.variables().items()})
        losses, step = [], 0
        dev = sess.device

        def feed(keys):
            nonlocal step
            inp = TI.step_inputs(model, rank, step)
            step += 1
            for k in keys:
                sess.stage_draws(k, inp[k])
            x = torch.tensor(inp['x'], device=dev)
            return (x, torch.tensor(inp['y'], device=dev)) if model == 'pix2pix' else x
        def snap(store):                                               # the all-reduced bucket holds the SUM over replicas
            if rank == 0:
                i = len([k for k in extra if k.startswith('nstep.')])
                extra['nstep.%d' % i] = np.zeros(0)
                extra.update({'grad.%d.%s' % (i, k.replace('/', '.')): v.detach().cpu().numpy() / world for k, v in store.grad_views.items()})
        for _ in range(TI.iterations(model)):
            if model == 'vae':
                rep.step(feed(['eps']))
                snap(rep.store)
                losses.append(rep.losses())
            elif model == 'pix2pix':
                rep.d_step(feed([]))
                snap(rep.d_store)
                rep.g_step(feed([]))
                snap(rep.g_store)
                losses.append(rep.report(feed([])))
            else:
                for _d in range(rep.args.n_disc_train):
                    rep.d_step(feed(['z', 'alpha']))
                    snap(rep.d_store)
                rep.g_step(feed(['z', 'alpha']))
                snap(rep.g_store)
                losses.append(rep.losses())
    else:
        losses = [rep.train_func() for _ in range(TI.ITERATIONS + 1)]          # eager, capture, 2 replays
    torch.cuda.synchronize()
    if rank == 0:
        _save(out_path, rep, losses, extra)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    model = sys.argv[2] if len(sys.argv) > 2 else 'iwgan'
    mode = sys.argv[3] if len(sys.argv) > 3 else 'same'
    if mode == 'towers':
        run_towers(sys.argv[1], model)
    else:
        run(sys.argv[1], int(os.environ.get('WORLD_SIZE', '1')), model, mode)
