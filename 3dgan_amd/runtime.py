"""Runtime context that plays the role of the reference's `sess` (tf.Session under
tf.train.Supervisor, train.py:254-273): device, replica identity, compute dtype, the
counter-based RNG streams that replace TF's unseeded Philox ops, and the gradient exchange
that replaces the CPU-side tower mean of util.py:118-147.

One process per GPU: rank r is tower r of the reference (util.py:54-77); weights and optimizer
state are replicated in every GPU's HBM, gradients are averaged with one RCCL all-reduce per
net on the flat f32 bucket (torch.distributed backend "nccl" = RCCL over xGMI).
"""
import os

import torch
import torch.distributed as dist

from . import _lib
from . import kernels as K


class Session:
    def __init__(self, device=None, dtype=K.BF16, seed=0, rank=None, world_size=None, check_numerics=False):
        if rank is None:
            rank = int(os.environ.get('RANK', '0'))
        if world_size is None:
            world_size = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank, self.world_size = rank, world_size
        if device is None:
            if not torch.cuda.is_available():
                raise _lib.TdgError('no MI355X visible: the 3dgan_amd hot path has no CPU fallback')
            device = local_device()
        self.device = torch.device(device)
        if self.device.type == 'cuda':
            torch.cuda.set_device(self.device)
        _lib.load()                                  # fail loudly now if the HIP library is missing
        self.dtype = dtype
        self.seed = int(seed)
        self.check_numerics = check_numerics
        self.global_step = 0                         # train.py:201 (bumped by BOTH apply ops, gan.py:79-81)
        self.global_epoch = 0                        # train.py:202
        self._draws = 0
        self._draws_dev = torch.zeros(1, dtype=torch.int32, device=self.device)   # same count, device resident
        self.inject = {}                             # tests: {'z': [..], 'alpha': [..]} consumed in order
        self.staged = {}                             # tests: {'z': device f32 tensor}, see stage_draws()
        self._flag = None

    # ---- RNG (tf.random_normal / tf.random_uniform, models/gan.py:246,224) ---------
    def _bump_draws(self):
        """The draw counter lives in device memory so that a captured hipGraph replays fresh streams; the RNG kernels
        count their own draw (tdg_random_*_dev), this is only the host's mirror of eagerly issued draws."""
        self._draws += 1

    def _injected(self, key):
        q = self.inject.get(key)
        if q:
            return q.pop(0)
        return None

    def stage_draws(self, key, values):
        """Tests: from now on every draw of site `key` ('z', 'alpha', 'eps', ...) is a device-to-device copy out of ONE
        fixed device buffer, which this call (re)fills from `values`.  Unlike `inject` (host arrays, eager steps only) the
        copy is capturable, so hipGraph-captured step bodies replay with injected draws: refill before every optimizer step."""
        t = torch.as_tensor(values, dtype=torch.float32).reshape(-1).to(self.device)
        buf = self.staged.get(key)
        if buf is None or buf.numel() != t.numel():
            if buf is not None:
                raise ValueError('staged draws of %r changed size (%d -> %d): the buffer address is captured' % (key, buf.numel(), t.numel()))
            self.staged[key] = t.clone()
        else:
            buf.copy_(t)

    def _staged_into(self, key, dst_flat, n):
        buf = self.staged.get(key)
        if buf is None:
            return False
        if buf.numel() < n:
            raise ValueError('staged draws of %r hold %d values, the site draws %d' % (key, buf.numel(), n))
        dst_flat[:n].copy_(buf[:n])
        return True

    def random_normal(self, act, n_rows, key='z'):
        """Fill the first n_rows images of `act` with N(0,1) (per-replica stream)."""
        inj = self._injected(key)
        n = n_rows * act.image_elems
        if inj is not None:
            t = torch.as_tensor(inj, dtype=torch.float32).reshape(-1)
            act.buf[:n].copy_(t.to(self.device, K.TORCH_DTYPE[act.dtype]))
            return
        if self._staged_into(key, act.buf, n):
            return
        _lib.call('tdg_random_normal_dev', act.dtype, self.seed, (self.rank << 8) | 1, K.ptr(self._draws_dev), n,
                  act.ptr(0), K.stream())
        self._bump_draws()

    def random_uniform(self, out, n, key='alpha'):
        inj = self._injected(key)
        if inj is not None:
            out[:n].copy_(torch.as_tensor(inj, dtype=torch.float32).reshape(-1).to(self.device))
            return
        if self._staged_into(key, out.view(-1), n):
            return
        _lib.call('tdg_random_uniform_f32_dev', self.seed, (self.rank << 8) | 2, K.ptr(self._draws_dev), n, K.ptr(out),
                  K.stream())
        self._bump_draws()

    # ---- gradient exchange (util.py:118-147 average_gradients) ---------------------------------------
    def allreduce_mean_scale(self, flat_grads):
        """Sum the flat bucket over replicas; returns the 1/n the optimizer kernel applies."""
        if self.world_size > 1:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
            return 1.0 / self.world_size
        return 1.0

    def allreduce_async(self, flat_slice):
        """Start summing a slice of a flat gradient bucket over replicas (RCCL runs on its own stream, after the work
        already enqueued on the current stream); returns a handle whose wait() orders the current stream behind it, or
        None on a single replica."""
        if self.world_size > 1:
            return dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, async_op=True)
        return None

    def allreduce_split(self, flat_grads, lo, hi, between=None):
        """The critic's exchange (models/gan.py d_step): the slice [lo, hi) of the flat bucket -- complete before the rest
        of the bucket is -- starts its all-reduce asynchronously, `between()` enqueues the work that finishes the rest
        (it runs underneath the exchange), then the remaining pieces follow.  Equal to ONE all-reduce of the whole
        bucket (tests/test_distributed_cpu.py).  Returns the 1/n the optimizer kernel applies."""
        work = self.allreduce_async(flat_grads[lo:hi])
        if between is not None:
            between()
        if work is not None:
            work.wait()
        if lo > 0:
            self.allreduce_mean_scale(flat_grads[:lo])
        if hi < flat_grads.numel():
            self.allreduce_mean_scale(flat_grads[hi:])
        return 1.0 / self.world_size

    def report_scalars(self, scal, mean=False):
        """The loss scalars every rank reports: those of the LAST replica (the reference's loss dict is overwritten tower
        by tower, util.py:187-193), or their mean over replicas with the opt-in --mean_loss."""
        if self.world_size <= 1 or not dist.is_initialized():
            return scal
        t = scal.clone()
        if mean:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t /= self.world_size
        else:
            dist.broadcast(t, src=dist.get_world_size() - 1)      # the last rank of the group actually joined
        return t

    def assert_finite(self, store, what):
        """--check_numerics (hem/util/training.py:52-53): name the offending variable.  With several replicas the flag is
        all-reduced (MAX) first, so every rank raises together instead of one rank leaving its peers in the next
        collective."""
        if not self.check_numerics:
            return
        flag = self._nonfinite_flag(store)
        if self.world_size > 1 and dist.is_initialized():
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            for name, gv in store.grad_views.items():
                if not bool(torch.isfinite(gv).all()):
                    raise FloatingPointError('%s: gradient of %s has NaN or Inf' % (what, name))
            raise FloatingPointError('%s: a gradient has NaN or Inf on another replica' % what)

    def _nonfinite_flag(self, store):
        """int32[1] on the device: 1 if the flat gradient bucket holds a NaN or Inf (one fused pass, tdg_check_finite)."""
        if self._flag is None:
            self._flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._flag.zero_()
        _lib.call('tdg_check_finite', K.ptr(store.grads), store.size, K.ptr(self._flag), K.stream())
        return self._flag

    # ---- RNG stream position (checkpoint / resume) ---------------------------------------------------------------
    def rng_state(self):
        """Number of z / alpha / eps / dropout draws made so far (the Philox counter offset of the next draw)."""
        return int(self._draws_dev.item())

    def set_rng_state(self, draws):
        self._draws = int(draws)
        self._draws_dev.fill_(int(draws))


def local_device():
    """cuda:<LOCAL_RANK>.  Only the one-GPU rehearsal of the N > 1 path (TDG_DIST_BACKEND=gloo) may place several ranks
    on one device; under RCCL a rank without a GPU of its own is an error, not a wrap-around."""
    n = torch.cuda.device_count()
    lr = int(os.environ.get('LOCAL_RANK', '0'))
    if lr >= max(n, 1):
        if os.environ.get('TDG_DIST_BACKEND') == 'gloo':
            return torch.device('cuda', lr % max(n, 1))
        raise _lib.TdgError('LOCAL_RANK %d >= %d visible GPUs (one process per GPU; set TDG_DIST_BACKEND=gloo only to '
                            'rehearse the multi-rank path on fewer devices)' % (lr, n))
    return torch.device('cuda', lr)


def init_distributed(backend=None):
    """Join the process group when launched by torch.distributed.run (RANK/WORLD_SIZE set): one process per GPU,
    backend "nccl" (= RCCL over xGMI); TDG_DIST_BACKEND=gloo rehearses the same code path without RCCL."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('TDG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if torch.cuda.is_available():
            torch.cuda.set_device(local_device())    # before the communicator is created
        dist.init_process_group(backend=backend)
    return ws


def broadcast_store(store, src=0):
    """Replicas start from identical variables (shared variables of the reference's towers)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(store.params, src=src)
