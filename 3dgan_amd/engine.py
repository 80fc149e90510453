"""Execution engine: binds the `Net`s recorded by ops/layers.py to HBM buffers and runs their
forward / backward / tangent passes as sequences of HIP kernels (include/tdg.h).

This replaces what TensorFlow does for the reference between graph construction and
`sess.run` (SURVEY.md L0 + the graph half of L1/L2): variable storage (`ParamStore`, one flat
f32 bucket per net so one RCCL all-reduce and one fused optimizer launch cover a whole net),
autodiff (hand-scheduled, including the second-order term of the gradient penalty), and the
optimizers of `util.py:150-183`.

Data layout in HBM: activations NHWC in the compute dtype with a channel stride
rounded up to 8 (zero padded); master weights, gradients and optimizer slots f32 in flat
buckets; each conv keeps two packed copies of its filter (forward / backward-data GEMM
operand layouts) that are refreshed after every optimizer step.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from . import kernels as K
from .ops import activations as A


# ------------------------------------------------------------------------------ parameters
class GraphRunner:
    """Mixin: run a step body eagerly the first time (lazy workspaces, kernel attributes), capture it into a hipGraph
    the second time and replay it afterwards.  Everything that varies per step must live at fixed device addresses
    (batch staging buffer, Philox draw counter, Adam step count), so that a replay IS a new step.  Collectives and
    host reads stay outside the captured bodies."""

    def init_graphs(self, args, sess):
        # (--check_numerics keeps the graphs: the finite check runs between the captured bodies, never inside one)
        self.use_graphs = bool(getattr(args, 'use_graphs', True)) and sess.device.type == 'cuda'
        self._warm, self._graphs = set(), {}

    def _run(self, name, body):
        if not self.use_graphs or self.sess.inject:
            return body()
        g = self._graphs.get(name)
        if g is not None:
            return g.replay()
        if name not in self._warm:
            self._warm.add(name)
            return body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread-local capture mode: the RCCL watchdog thread of a multi-rank run polls events while this thread captures
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            body()
        self._graphs[name] = g
        g.replay()


class ParamStore:
    """Flat f32 parameter / gradient buckets of one net with named views
    (names as in the reference: `generator/vars/fc1/weights`, `generator/BatchNorm/beta`)."""

    def __init__(self, device):
        self.device = device
        self.index = {}            # name -> (offset, shape)
        self.size = 0
        self.params = self.grads = None
        self.views, self.grad_views = {}, {}

    def declare(self, name, shape):
        if name in self.index:
            raise ValueError('variable %s already exists' % name)
        n = int(np.prod(shape))
        self.index[name] = (self.size, tuple(shape))
        self.size += (n + 3) // 4 * 4                    # keep every variable 16-byte aligned

    def allocate(self):
        self.params = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        for name, (off, shape) in self.index.items():
            n = int(np.prod(shape))
            self.views[name] = self.params[off:off + n].view(shape)
            self.grad_views[name] = self.grads[off:off + n].view(shape)

    def __getitem__(self, name):
        return self.views[name]

    def grad(self, name):
        return self.grad_views[name]

    def load(self, arrays):
        for name in self.index:
            self.views[name].copy_(torch.as_tensor(np.asarray(arrays[name]), dtype=torch.float32))

    def state_dict(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.views.items()}

    def grads_dict(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.grad_views.items()}

    def num_params(self):
        return sum(int(np.prod(s)) for _, s in self.index.values())


def xavier_uniform_(t, shape, gen):
    """tf.contrib.layers.xavier_initializer(), also used for biases (ops/layers.py:52-53; App. A-4)."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    t.copy_((torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * lim)


# ------------------------------------------------------------------------------ optimizers
class Optimizer:
    """TF-1.x update rules on a flat bucket; `grad_scale` folds the
    1/n_replicas of `average_gradients` (util.py:138-139) into the fused step."""

    def __init__(self, store):
        self.store = store
        self.t = 0

    def _slot(self, fill=0.0):
        return torch.full_like(self.store.params, fill)

    def set_step_count(self, t):
        self.t = int(t)

    def state_tensors(self):
        return {}


class Adam(Optimizer):
    def __init__(self, store, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        super().__init__(store)
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m, self.v = self._slot(), self._slot()
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=store.device)    # step count for graph replay

    @property
    def t(self):
        return int(self.t_dev.item())             # the device counter is authoritative (graph replays bump only it)

    @t.setter
    def t(self, value):
        if hasattr(self, 't_dev'):
            self.t_dev.fill_(int(value))

    def step(self, grad_scale=1.0):
        """lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is derived in-kernel from the device-resident step count."""
        s = self.store
        _lib.call('tdg_adam_step_dev', K.ptr(s.params), K.ptr(s.grads), K.ptr(self.m), K.ptr(self.v), s.size,
                  self.lr, self.b1, self.b2, self.eps, grad_scale, K.ptr(self.t_dev), K.stream())     # counts the step itself

    def set_step_count(self, t):
        self.t_dev.fill_(int(t))

    def state_tensors(self):
        return {'m': self.m, 'v': self.v}


class RMSProp(Optimizer):
    """tf.train.RMSPropOptimizer (util.py:161-164), including `--centered`."""

    def __init__(self, store, lr, decay=0.9, momentum=0.0, eps=1e-10, centered=False):
        super().__init__(store)
        self.lr, self.decay, self.mu, self.eps, self.centered = lr, decay, momentum, eps, bool(centered)
        self.rms, self.mom = self._slot(1.0), self._slot()        # rms slot starts at ONE in TF
        self.mg = self._slot() if self.centered else None

    def step(self, grad_scale=1.0):
        self.t += 1
        s = self.store
        if self.centered:
            _lib.call('tdg_rmsprop_centered_step', K.ptr(s.params), K.ptr(s.grads), K.ptr(self.mg), K.ptr(self.rms),
                      K.ptr(self.mom), s.size, self.lr, self.decay, self.mu, self.eps, grad_scale, K.stream())
        else:
            _lib.call('tdg_rmsprop_step', K.ptr(s.params), K.ptr(s.grads), K.ptr(self.rms), K.ptr(self.mom), s.size,
                      self.lr, self.decay, self.mu, self.eps, grad_scale, K.stream())

    def state_tensors(self):
        d = {'rms': self.rms, 'mom': self.mom}
        if self.centered:
            d['mg'] = self.mg
        return d


class Adagrad(Optimizer):
    """tf.train.AdagradOptimizer (util.py:167-168); ProximalAdagrad with its default zero l1/l2 strengths
    (util.py:173-174) is the same update."""

    def __init__(self, store, lr, initial_accumulator_value=0.1):
        super().__init__(store)
        self.lr = lr
        self.acc = self._slot(initial_accumulator_value)

    def step(self, grad_scale=1.0):
        self.t += 1
        s = self.store
        _lib.call('tdg_adagrad_step', K.ptr(s.params), K.ptr(s.grads), K.ptr(self.acc), s.size, self.lr, grad_scale,
                  K.stream())

    def state_tensors(self):
        return {'acc': self.acc}


class Adadelta(Optimizer):
    """tf.train.AdadeltaOptimizer (util.py:165-166): rho 0.95, eps 1e-8."""

    def __init__(self, store, lr, rho=0.95, eps=1e-8):
        super().__init__(store)
        self.lr, self.rho, self.eps = lr, rho, eps
        self.acc, self.acc_update = self._slot(), self._slot()

    def step(self, grad_scale=1.0):
        self.t += 1
        s = self.store
        _lib.call('tdg_adadelta_step', K.ptr(s.params), K.ptr(s.grads), K.ptr(self.acc), K.ptr(self.acc_update), s.size,
                  self.lr, self.rho, self.eps, grad_scale, K.stream())

    def state_tensors(self):
        return {'acc': self.acc, 'acc_update': self.acc_update}


class Ftrl(Optimizer):
    """tf.train.FtrlOptimizer (util.py:182-183): lr_power -0.5, accumulator 0.1, l1 = l2 = 0."""

    def __init__(self, store, lr, initial_accumulator_value=0.1, l1=0.0, l2=0.0):
        super().__init__(store)
        self.lr, self.l1, self.l2 = lr, l1, l2
        self.acc, self.linear = self._slot(initial_accumulator_value), self._slot()

    def step(self, grad_scale=1.0):
        self.t += 1
        s = self.store
        _lib.call('tdg_ftrl_step', K.ptr(s.params), K.ptr(s.grads), K.ptr(self.acc), K.ptr(self.linear), s.size,
                  self.lr, self.l1, self.l2, grad_scale, K.stream())

    def state_tensors(self):
        return {'acc': self.acc, 'linear': self.linear}


class Momentum(Optimizer):
    def __init__(self, store, lr, momentum=0.0):
        super().__init__(store)
        self.lr, self.mu = lr, momentum
        self.acc = self._slot()

    def step(self, grad_scale=1.0):
        self.t += 1
        s = self.store
        _lib.call('tdg_sgd_momentum_step', K.ptr(s.params), K.ptr(s.grads), K.ptr(self.acc), s.size,
                  self.lr, self.mu, grad_scale, K.stream())

    def state_tensors(self):
        return {'acc': self.acc}


# ------------------------------------------------------------------------------ sequential net
def _alias(act, h, w, c):
    """Same storage, different logical NHWC shape (tf.reshape); needs an unpadded layout."""
    if (act.h, act.w, act.c) == (h, w, c):
        return act
    if act.cs != act.c or act.h * act.w * act.c != h * w * c or K.pad_channels(c) != c:
        raise ValueError('reshape between layers needs unpadded channel layouts (%s -> %s)' %
                         ((act.h, act.w, act.c, act.cs), (h, w, c)))
    return K.Act(act.n, h, w, c, act.dtype, act.buf.device, c, act.buf)


class BoundLayer:
    pass


class SeqNet:
    """A chain of dense/conv2d/deconv2d layers bound to buffers for `capacity` images.

    Gradient bookkeeping per layer i:  delta[i] = dL/d(conv output incl. bias);
    gout[i] = dL/d(layer output after BN/activation) when that is a separate tensor
    (layers with BN, tanh or sigmoid); otherwise the consumer's backward-data epilogue
    multiplies by the (l)relu derivative and writes delta[i] directly.
    """

    def __init__(self, net, capacity, in_shape, dtype, device, store, n_bn_passes=1, need_input_grad=False,
                 tangent_capacity=0, ws=None, out_act=None, out_grad=None, forward_only=False):
        """forward_only: no gradient buffers are bound (a net that only ever runs forward / forward_groups)."""
        self.net, self.cap, self.dtype, self.device, self.store = net, capacity, dtype, device, store
        self.forward_only = forward_only
        if forward_only and (need_input_grad or tangent_capacity or out_grad is not None):
            raise ValueError('a forward-only net has no gradient buffers')
        self.ws = ws or K.Workspace(device)
        self._pack_jobs = None
        self.n_bn_passes = n_bn_passes
        h, w, c = in_shape
        self.x = K.Act(capacity, h, w, c, dtype, device)                   # net input
        self.dx = K.Act(capacity, h, w, c, dtype, device) if need_input_grad else None
        self.layers = []
        prev = self.x
        for spec in net.layers:
            if getattr(spec, 'dropout', 0):
                raise NotImplementedError('layer %s: dropout is only executed by the pix2pix U-Net (models/pix2pix.py)' % spec.name)
        prev_g = self.dx
        cum = 1
        for idx, spec in enumerate(net.layers):
            L = BoundLayer()
            L.spec, L.idx = spec, idx
            ih, iw, ic = spec.in_shape
            oh, ow, oc = spec.out_shape
            # one dense "image row" may be a fraction of an image
            local = (prev.h * prev.w * prev.c) // (ih * iw * ic)
            cum *= local
            rpi = L.rpi = cum                        # rows of this layer per original image
            L.inp = _alias_rows(prev, local, ih, iw, ic)
            L.gin = _alias_rows(prev_g, local, ih, iw, ic) if prev_g is not None else None
            L.act = spec.act if spec.act is not None else A.identity
            L.rowdot = spec.kind == 'dense' and spec.out_size == 1
            L.wname, L.bname = net.var_name(spec, 'weights'), net.var_name(spec, 'bias')
            if L.rowdot:
                L.out = torch.zeros(capacity * rpi, dtype=torch.float32, device=device)       # f32 scores
                L.seed = torch.zeros(capacity * rpi, dtype=torch.float32, device=device)      # dL/d(score)
                L.h = L.pre = L.delta = L.gout = None
            else:
                last = idx == len(net.layers) - 1
                L.h = out_act if (last and out_act is not None) else K.Act(capacity * rpi, oh, ow, oc, dtype, device)
                if (L.h.n, L.h.h, L.h.w, L.h.c) != (capacity * rpi, oh, ow, oc):
                    raise ValueError('out_act does not match the last layer output')
                L.pre = L.h.like() if spec.normed and spec.kind != 'residual' else None       # batch norm: the normalised pre-activation; instance norm: the conv output
                L.delta = L.h.like() if not forward_only else None
                sep = spec.normed or spec.kind == 'residual' or L.act.code in (K.ACT_TANH, K.ACT_SIGMOID)
                L.gout = (L.h.like() if sep else L.delta) if not forward_only else None
                if last and out_grad is not None:
                    if not sep:
                        L.delta = out_grad
                    L.gout = out_grad
                if spec.use_bn and spec.kind != 'residual':
                    L.bn_stats = [torch.zeros(2 * oc, dtype=torch.float32, device=device) for _ in range(n_bn_passes)]
                    L.bn_names = [net.bn_name(p, idx) for p in range(n_bn_passes)]
                if spec.use_in:
                    L.in_stats = torch.zeros(capacity * rpi * 2 * oc, dtype=torch.float32, device=device)
                    L.in_names = (net.var_name(spec, 'scale'), net.var_name(spec, 'shift'))
                if (spec.use_in or spec.kind == 'residual') and tangent_capacity:
                    raise NotImplementedError('layer %s: no gradient-penalty tangent pass through instance norm / residual blocks' % spec.name)
                if spec.kind == 'residual':
                    L.res = Residual(self, L, net, idx, capacity * rpi, n_bn_passes)
                    L.conv = L.tan = L.pre = None
                    self.layers.append(L)
                    prev, prev_g = L.h, L.gout
                    continue
                if spec.kind == 'deconv2d':
                    big, small = L.h, L.inp
                else:                                  # conv2d, dense, and conv A of a residual block
                    big, small = L.inp, L.h
                if spec.padding == 'SAME':
                    pt = max((small.h - 1) * spec.stride + spec.k - big.h, 0) // 2
                    pl = max((small.w - 1) * spec.stride + spec.k - big.w, 0) // 2
                else:
                    pt = pl = 0
                L.conv = K.Conv(big, small, spec.k, spec.k, spec.stride, pt, pl)
                L.tan = K.Act(tangent_capacity * rpi, oh, ow, oc, dtype, device) if tangent_capacity else None
            self.layers.append(L)
            prev, prev_g = (L.h, L.gout) if not L.rowdot else (None, None)
        self.tangent_capacity = tangent_capacity
        if tangent_capacity:
            self.tan_in = K.Act(tangent_capacity, h, w, c, dtype, device)

    # -- variables ---------------------------------------------------------------------------------
    def declare_variables(self):
        for L in self.layers:
            if L.spec.kind == 'residual':
                L.res.declare(self.store)
                continue
            self.store.declare(L.wname, L.spec.filter_shape)
            self.store.declare(L.bname, (L.spec.out_size,))
            if L.spec.use_in:
                for name in L.in_names:
                    self.store.declare(name, (L.spec.out_size,))
        for p in range(self.n_bn_passes):
            for L in self.layers:
                if L.spec.kind == 'residual':
                    L.res.declare_bn(self.store, p)
                elif L.spec.use_bn:
                    self.store.declare(L.bn_names[p], (L.spec.out_size,))

    def init_variables(self, gen):
        """Fresh variables: xavier-uniform weights and biases, zero betas (App. A-3/A-4);
        'normal0.02' (pix2pix, hem/models/pix2pix.py:180) draws N(0, 0.02)."""
        for L in self.layers:
            if L.spec.use_in:
                self.store[L.in_names[0]].fill_(1.0)        # scale: ones_initializer; shift stays at zeros (hem/ops/images.py:79-80)
            pairs = L.res.variables() if L.spec.kind == 'residual' else ((L.wname, L.spec.filter_shape), (L.bname, (L.spec.out_size,)))
            for name, shape in pairs:
                cpu = torch.empty(shape, dtype=torch.float32)
                if L.spec.init == 'xavier':
                    xavier_uniform_(cpu, shape, gen)
                else:
                    cpu.copy_(torch.randn(shape, generator=gen) * 0.02)
                self.store[name].copy_(cpu)

    def repack(self):
        """Refresh the packed GEMM operands from the f32 masters (after every optimizer step)."""
        if self._pack_jobs is None:
            # masters and packed buffers never move, so the job table is built once
            jl = []
            for L in self.layers:
                if L.spec.kind == 'residual':
                    jl += L.res.pack_jobs(self.store)
                elif not L.rowdot:
                    jl.append(L.conv.pack_job(self.store[L.wname]))
            self._pack_jobs = K.make_pack_jobs(jl) if jl else ()
        if len(self._pack_jobs):
            K.pack_all(self._pack_jobs)

    def share_filters(self, other):
        """Use `other`'s packed GEMM operands (a second binding of the same variables at another capacity): the packed
        layouts depend on the filter geometry only, so neither net packs twice and this one never repacks."""
        for L, O in zip(self.layers, other.layers):
            if L.rowdot:
                continue
            if L.spec.kind == 'residual' or L.spec is not O.spec:
                raise NotImplementedError('share_filters: layer %s' % L.spec.name)
            if (L.conv.fwd_bytes, L.conv.bwd_bytes) != (O.conv.fwd_bytes, O.conv.bwd_bytes):
                raise ValueError('layer %s: packed filter sizes differ between the two bindings' % L.spec.name)
            L.conv.w_fwd, L.conv.w_bwd = O.conv.w_fwd, O.conv.w_bwd
        self._pack_jobs = ()

    # -- forward -----------------------------------------------------------------------------------
    def forward_groups(self, ngroups, n):
        """forward(keep_pre=False) of `ngroups` batches of n images that lie behind each other (images [0, ngroups * n)):
        every conv is ONE GEMM over all of them, every batch norm takes its statistics per batch (tdg_bn_fwd_groups) --
        what ngroups separate forward() calls compute, with the GEMMs at ngroups times the rows."""
        if ngroups * n > self.cap:
            raise ValueError('%d x %d images, capacity %d' % (ngroups, n, self.cap))
        for L in self.layers:
            if L.rowdot or L.spec.kind == 'residual' or L.spec.use_in:
                raise NotImplementedError('forward_groups: layer %s' % L.spec.name)
            rn = ngroups * n * L.rpi
            bias = self.store[L.bname]
            if L.spec.use_bn:
                epi, target = K.epilogue(bias=bias), L.pre
            else:
                epi, target = K.epilogue(bias=bias, act=L.act.code, leak=L.act.leak), L.h
            if L.spec.kind == 'deconv2d':
                L.conv.bwd_data(L.inp.ptr(0), target.ptr(0), rn, epi)
            else:
                L.conv.fwd(L.inp.ptr(0), target.ptr(0), rn, epi)
            if L.spec.use_bn:
                C = L.spec.out_size
                if getattr(L, 'bn_stats_groups', None) is None or L.bn_stats_groups.numel() < ngroups * 2 * C:
                    L.bn_stats_groups = torch.zeros(ngroups * 2 * C, dtype=torch.float32, device=self.device)
                K.bn_fwd_groups(self.ws, L.pre, C, self.store[L.bn_names[0]], L.act.code, None, L.h, L.bn_stats_groups,
                                n * L.rpi * L.h.h * L.h.w, ngroups, leak=L.act.leak)
        return self.layers[-1].h

    def forward(self, img0, n, bn_pass=0, keep_pre=True):
        """Layers on images [img0, img0+n).  Returns the last layer's output (Act or f32 scores).  keep_pre=False: no backward
        pass will follow, so batch-norm layers skip writing their normalised pre-activation (only its activation)."""
        for L in self.layers:
            r0, rn = img0 * L.rpi, n * L.rpi
            if L.rowdot:
                cols = L.spec.in_size
                _lib.call('tdg_rowdot', self.dtype, L.inp.ptr(r0), rn, cols, K.ptr(self.store[L.wname]),
                          K.ptr(self.store[L.bname]), L.act.code, K.ptr(L.out, 4 * r0), K.stream())
                continue
            if L.spec.kind == 'residual':
                L.res.forward(r0, rn, bn_pass)
                continue
            bias = self.store[L.bname]
            if L.spec.use_in:
                epi = K.epilogue(bias=bias)
                target = L.pre
            elif L.spec.use_bn:
                # the GEMM's epilogue also emits the batch statistics' column partials of the tile it stores
                epi = K.colsum_epilogue(self.ws, rn * L.h.h * L.h.w, L.spec.out_size, K.COL_BN, bias=bias)
                target = L.pre
            else:
                epi = K.epilogue(bias=bias, act=L.act.code, leak=L.act.leak)
                target = L.h
            if L.spec.kind == 'deconv2d':
                L.conv.bwd_data(L.inp.ptr(r0), target.ptr(r0), rn, epi)
            else:
                L.conv.fwd(L.inp.ptr(r0), target.ptr(r0), rn, epi)
            if L.spec.use_in:
                K.in_fwd(L.pre, rn, L.spec.out_size, self.store[L.in_names[0]], self.store[L.in_names[1]], L.act.code, L.h,
                         L.in_stats[r0 * 2 * L.spec.out_size:], leak=L.act.leak, u_ptr=L.pre.ptr(r0), h_ptr=L.h.ptr(r0))
            elif L.spec.use_bn:
                rows = rn * L.h.h * L.h.w
                if K.nblk(epi):
                    K.bn_fwd_from_partials(epi, L.pre, L.spec.out_size, self.store[L.bn_names[bn_pass]], L.act.code,
                                           L.pre if keep_pre else None, L.h,
                                           L.bn_stats[bn_pass], bias, rows=rows, leak=L.act.leak,
                                           u_ptr=L.pre.ptr(r0), pre_ptr=L.pre.ptr(r0), h_ptr=L.h.ptr(r0))
                else:                                  # (f32 tiles, thin layers: the separate statistics pass)
                    K.bn_fwd(self.ws, L.pre, L.spec.out_size, self.store[L.bn_names[bn_pass]], L.act.code, L.pre, L.h,
                             L.bn_stats[bn_pass], rows=rows, leak=L.act.leak,
                             u_ptr=L.pre.ptr(r0), pre_ptr=L.pre.ptr(r0), h_ptr=L.h.ptr(r0))
        last = self.layers[-1]
        return last.out if last.rowdot else last.h

    # -- backward ----------------------------------------------------------------------------------
    def backward(self, img0, n, bn_pass=0, want_params=True, want_dx=False, acc=False, param_images=None,
                 dx_images=None, defer_wgrad=False):
        """Backprop from the last layer's seed (rowdot: L.seed; else L.gout) on images
        [img0, img0+n).  Parameter gradients cover `param_images` = (first, count) (default: the same
        range) and are accumulated into the store when `acc`.  With `defer_wgrad` the conv filter gradients are left
        to a later merged_wgrad() (bias gradients are still taken here).  Returns dL/d(net input) if asked."""
        beta = 1.0 if acc else 0.0
        p0, pn = (img0, n) if param_images is None else param_images
        g = self.store.grad
        have_db = None          # (layer, epilogue) whose backward-data GEMM has just emitted that layer's bias-gradient partials
        for L in reversed(self.layers):
            r0, rn = img0 * L.rpi, n * L.rpi
            q0, qn = p0 * L.rpi, pn * L.rpi
            below = self.layers[L.idx - 1] if L.idx > 0 else None
            need_in = below is not None or want_dx
            d0, dn, dimg = r0, rn, img0
            if below is None and want_dx and dx_images is not None:      # restrict the input gradient
                d0, dn, dimg = dx_images[0] * L.rpi, dx_images[1] * L.rpi, dx_images[0]
            if below is not None and not below.spec.normed and below.spec.kind != 'residual' and below.act.code in (K.ACT_LRELU, K.ACT_RELU):
                mmode = K.MASK_LRELU if below.act.code == K.ACT_LRELU else K.MASK_RELU
                msrc, mleak = below.h, below.act.leak
            else:
                mmode, msrc, mleak = K.MASK_NONE, None, 0.0
            if L.rowdot:
                cols = L.spec.in_size
                if want_params and qn > 0:
                    K.colsum_weighted(self.ws, self.dtype, L.inp.ptr(q0), qn, cols, cols, _sub(L.seed, q0, qn),
                                      g(L.wname), beta)
                    _lib.call('tdg_sum_f32', K.ptr(L.seed, 4 * q0), qn, K.ptr(g(L.bname)), beta, K.stream())      # (one launch)
                if need_in:
                    _lib.call('tdg_rowouter', self.dtype, K.ptr(L.seed, 4 * d0), K.ptr(self.store[L.wname]), dn, cols,
                              mmode, mleak, msrc.ptr(dimg * below.rpi) if msrc is not None else None, L.gin.ptr(d0), K.stream())
                continue
            if L.spec.kind == 'residual':
                L.res.backward(r0, rn, q0, qn, bn_pass, want_params, beta, need_in, mmode, mleak,
                               msrc.ptr(dimg * below.rpi) if msrc is not None else None)
                have_db = None
                continue
            # dL/d(conv output)
            hw = L.h.h * L.h.w
            if L.spec.use_in:
                scale, shift = self.store[L.in_names[0]], self.store[L.in_names[1]]
                if want_params:
                    dsc, dsh = g(L.in_names[0]), g(L.in_names[1])
                else:
                    if getattr(L, 'in_sink', None) is None:
                        L.in_sink = torch.zeros(2 * L.spec.out_size, dtype=torch.float32, device=self.device)
                    dsc, dsh = L.in_sink[:L.spec.out_size], L.in_sink[L.spec.out_size:]
                K.in_bwd(self.ws, L.gout, L.pre, rn, L.spec.out_size, scale, shift, L.in_stats[r0 * 2 * L.spec.out_size:], L.act.code,
                         L.delta, dsc, dsh, leak=L.act.leak, beta=beta if want_params else 0.0,
                         dh_ptr=L.gout.ptr(r0), u_ptr=L.pre.ptr(r0), du_ptr=L.delta.ptr(r0))
            elif L.spec.use_bn:
                if want_params:
                    dbeta = g(L.bn_names[bn_pass])
                else:                                   # keep the stored beta gradient of an earlier pass intact
                    if getattr(L, 'dbeta_sink', None) is None:
                        L.dbeta_sink = torch.zeros(L.spec.out_size, dtype=torch.float32, device=self.device)
                    dbeta = L.dbeta_sink
                # the pass that writes delta also sums its columns: the conv's bias gradient, when it covers the same rows
                db_here = want_params and (q0, qn) == (r0, rn)
                K.bn_bwd(self.ws, L.gout, L.pre, L.spec.out_size, self.store[L.bn_names[bn_pass]], L.bn_stats[bn_pass],
                         L.act.code, L.delta, dbeta, rows=rn * hw, leak=L.act.leak, beta_acc=0.0,
                         dh_ptr=L.gout.ptr(r0), pre_ptr=L.pre.ptr(r0), du_ptr=L.delta.ptr(r0),
                         dbias=g(L.bname) if db_here else None, dbias_acc=beta)
                if db_here:
                    have_db = (L, None)
            elif L.act.code in (K.ACT_TANH, K.ACT_SIGMOID):
                _lib.call('tdg_act_bwd', self.dtype, L.gout.ptr(r0), L.h.ptr(r0), rn * L.h.image_elems, L.act.code,
                          L.act.leak, L.delta.ptr(r0), K.stream())
            elif L.idx == len(self.layers) - 1 and L.act.code in (K.ACT_LRELU, K.ACT_RELU):
                _lib.call('tdg_act_bwd', self.dtype, L.gout.ptr(r0), L.h.ptr(r0), rn * L.h.image_elems, L.act.code,
                          L.act.leak, L.delta.ptr(r0), K.stream())
            if want_params and qn > 0:
                if have_db is not None and have_db[0] is L and have_db[1] is None:
                    pass                                   # taken by bn_bwd above
                elif have_db is not None and have_db[0] is L and K.nblk(have_db[1]):
                    K.bias_grad_from_partials(have_db[1], L.spec.out_size, g(L.bname), beta)
                else:
                    K.bias_grad(self.ws, L.delta, L.spec.out_size, g(L.bname), rows=qn * hw, beta=beta, dy_ptr=L.delta.ptr(q0))
                if defer_wgrad:
                    pass
                elif L.spec.kind == 'deconv2d':
                    L.conv.bwd_filter(L.delta.ptr(q0), L.inp.ptr(q0), g(L.wname), qn, beta)
                else:
                    L.conv.bwd_filter(L.inp.ptr(q0), L.delta.ptr(q0), g(L.wname), qn, beta)
            if need_in:
                mptr = msrc.ptr(dimg * below.rpi) if msrc is not None else None
                # what this GEMM stores IS delta of the layer below when that layer has no batch norm and a (l)relu / identity
                # activation: its bias gradient = the column sums of the stored tiles, restricted to the parameter images
                fuse = (want_params and below is not None and not below.rowdot and not below.spec.normed and below.spec.kind != 'residual' and
                        below.act.code in (K.ACT_LRELU, K.ACT_RELU, K.ACT_NONE) and below.gout is below.delta and
                        L.rpi == 1 and below.rpi == 1 and p0 == img0 and 0 < pn <= n and (d0, dn) == (r0, rn))
                if fuse:
                    epi = K.colsum_epilogue(self.ws, dn * below.h.h * below.h.w, below.spec.out_size, K.COL_SUM,
                                            images=(pn if pn < n else 0), mask_mode=mmode, leak=mleak, mask_src=mptr)
                    have_db = (below, epi)
                else:
                    epi = K.epilogue(mask_mode=mmode, leak=mleak, mask_src=mptr)
                    have_db = None
                if L.spec.kind == 'deconv2d':
                    L.conv.fwd(L.delta.ptr(d0), L.gin.ptr(d0), dn, epi)
                else:
                    L.conv.bwd_data(L.delta.ptr(d0), L.gin.ptr(d0), dn, epi)
        return self.dx

    # -- gradient-penalty tangent pass (models/gan.py:228 second order; oracle/gan_ref.py) -------
    def tangent_forward(self, img0, n, acc=True):
        """With u in `self.tan_in` (n images): the tangents t_i = act'(h_i) * conv_i(t_{i-1}) (no bias) of every conv
        layer, and d(fc2 weights) += sum_rows t_last.  Only BN-free (l)relu critics (the iwgan D)."""
        beta = 1.0 if acc else 0.0
        g = self.store.grad
        t_prev = self.tan_in
        for L in self.layers:
            r0, rn = img0 * L.rpi, n * L.rpi
            if L.rowdot:
                cols = L.spec.in_size
                K.colsum_weighted(self.ws, self.dtype, t_prev.ptr(0), rn, cols, cols, None, g(L.wname), beta)
                break
            if L.spec.normed or L.spec.kind != 'conv2d' or L.act.code not in (K.ACT_LRELU, K.ACT_RELU):
                raise NotImplementedError('tangent pass: layer %s is not a BN-free (l)relu conv' % L.spec.name)
            L.tan_src = _alias_rows(t_prev, L.rpi, *L.spec.in_shape) if t_prev is not self.tan_in else self.tan_in
            mmode = K.MASK_LRELU if L.act.code == K.ACT_LRELU else K.MASK_RELU
            L.conv.fwd(L.tan_src.ptr(0), L.tan.ptr(0), rn, K.epilogue(mask_mode=mmode, leak=L.act.leak, mask_src=L.h.ptr(r0)))
            t_prev = L.tan

    def tangent_wgrad(self, img0, n, layers, acc=True):
        """dW_i += bwd_filter(t_{i-1}, delta_i) for the given conv layers (after tangent_forward, with the first-order
        deltas of images [img0, img0+n) in place).  The layers are independent of each other: the caller picks the
        order (the largest filter first, so that its all-reduce can start while the others are still being computed)."""
        beta = 1.0 if acc else 0.0
        for L in layers:
            L.conv.bwd_filter(L.tan_src.ptr(0), L.delta.ptr(img0 * L.rpi), self.store.grad(L.wname), n * L.rpi, beta)

    def conv_layers(self):
        return [L for L in self.layers if not L.rowdot]

    def merged_wgrad(self, n_first, n_tan, layers, acc=False):
        """dW_i (+)= bwd_filter over images [0, n_first) of the layer input (first-order rows) AND the n_tan tangent rows
        t_{i-1} against delta_i of images [0, n_first + n_tan) -- ONE GEMM per layer (tdg_conv2d_bwd_filter2) where the
        first-order and the tangent-pass filter gradients used to be two plus two slab reductions.  Needs the deltas of
        backward(0, n_first + n_tan, ..., defer_wgrad=True) and tangent_forward(n_first, n_tan) in place."""
        beta = 1.0 if acc else 0.0
        for L in layers:
            L.conv.bwd_filter2(L.inp.ptr(0), n_first * L.rpi, L.tan_src.ptr(0), L.delta.ptr(0), self.store.grad(L.wname),
                               (n_first + n_tan) * L.rpi, beta)

    def tangent_backward(self, img0, n, acc=True):
        """The whole tangent pass, layer by layer: dW_i += bwd_filter(t_{i-1}, delta_i) right after t_{i-1} was produced
        (it is still cache-resident then: measured 0.14 ms per D step faster than tangents first, filter gradients after)."""
        beta = 1.0 if acc else 0.0
        g = self.store.grad
        t_prev = self.tan_in
        for L in self.layers:
            r0, rn = img0 * L.rpi, n * L.rpi
            if L.rowdot:
                cols = L.spec.in_size
                K.colsum_weighted(self.ws, self.dtype, t_prev.ptr(0), rn, cols, cols, None, g(L.wname), beta)
                break
            if L.spec.normed or L.spec.kind != 'conv2d' or L.act.code not in (K.ACT_LRELU, K.ACT_RELU):
                raise NotImplementedError('tangent pass: layer %s is not a BN-free (l)relu conv' % L.spec.name)
            t_in = _alias_rows(t_prev, L.rpi, *L.spec.in_shape) if t_prev is not self.tan_in else self.tan_in
            L.conv.bwd_filter(t_in.ptr(0), L.delta.ptr(r0), g(L.wname), rn, beta)
            mmode = K.MASK_LRELU if L.act.code == K.ACT_LRELU else K.MASK_RELU
            L.conv.fwd(t_in.ptr(0), L.tan.ptr(0), rn, K.epilogue(mask_mode=mmode, leak=L.act.leak, mask_src=L.h.ptr(r0)))
            t_prev = L.tan


class Residual:
    """The two-conv residual block of the gen-2 layer surface (hem/ops/layers.py:215-320) as one SeqNet layer:
        S = conv_A(x) + b_A (`shortcut`);  hA = act([bn](S));  T = conv_B(hA) + b_B;  out = act([bn](T) + S).
    The block's output and incoming gradient are separate tensors (L.h / L.gout: the consumer does not apply this layer's
    activation derivative); its own backward delivers dL/dx with the derivative mask of the layer below, like any conv."""

    def __init__(self, seq, L, net, idx, capacity, n_bn_passes):
        self.seq, self.L, self.net, self.idx = seq, L, net, idx
        spec = self.spec = L.spec
        dt, dev = seq.dtype, seq.device
        oh, ow, oc = spec.out_shape
        A = lambda: K.Act(capacity, oh, ow, oc, dt, dev)
        self.S, self.hA, self.T = A(), A(), A()                       # shortcut, first activation, conv B output
        self.dZ, self.dT, self.dhA, self.dS = A(), A(), A(), A()
        self.preA = A() if spec.use_bn else None                      # batch-normalised S / T (the activations' inputs)
        self.preB = A() if spec.use_bn else None
        pt = max((oh - 1) + spec.k - spec.in_shape[0], 0) // 2
        pl = max((ow - 1) + spec.k - spec.in_shape[1], 0) // 2
        self.convA = K.Conv(L.inp, self.S, spec.k, spec.k, 1, pt, pl)
        self.convB = K.Conv(self.hA, self.T, spec.k, spec.k, 1, pt, pl)
        self.names = {w: '%s/vars/%s%s/%s' % (net.name, spec.name, ab, w2) for ab in 'AB' for w2 in ('weights', 'bias')
                      for w in ['%s%s' % (ab, w2[0])]}              # Aw, Ab, Bw, Bb
        self.shapes = {'Aw': (spec.k, spec.k, spec.in_size, oc), 'Ab': (oc,), 'Bw': (spec.k, spec.k, oc, oc), 'Bb': (oc,)}
        self._sink = torch.zeros(oc, dtype=torch.float32, device=dev)     # beta gradients of passes that do not want parameters
        if spec.use_bn:
            self.statsA = [torch.zeros(2 * oc, dtype=torch.float32, device=dev) for _ in range(n_bn_passes)]
            self.statsB = [torch.zeros(2 * oc, dtype=torch.float32, device=dev) for _ in range(n_bn_passes)]
            self.bnA = [net.bn_name(p, idx, 0) for p in range(n_bn_passes)]
            self.bnB = [net.bn_name(p, idx, 1) for p in range(n_bn_passes)]

    def declare(self, store):
        for k in ('Aw', 'Ab', 'Bw', 'Bb'):
            store.declare(self.names[k], self.shapes[k])

    def declare_bn(self, store, p):
        if self.spec.use_bn:
            store.declare(self.bnA[p], (self.spec.out_size,))
            store.declare(self.bnB[p], (self.spec.out_size,))

    def variables(self):
        return [(self.names[k], self.shapes[k]) for k in ('Aw', 'Ab', 'Bw', 'Bb')]

    def pack_jobs(self, store):
        return [self.convA.pack_job(store[self.names['Aw']]), self.convB.pack_job(store[self.names['Bw']])]

    def forward(self, r0, rn, bn_pass):
        st, sp, L, ws = self.seq.store, self.spec, self.L, self.seq.ws
        act, oc, dt = L.act, sp.out_size, self.seq.dtype
        rows = rn * self.S.h * self.S.w
        nel = rn * self.S.image_elems
        self.convA.fwd(L.inp.ptr(r0), self.S.ptr(r0), rn, K.epilogue(bias=st[self.names['Ab']]))
        if sp.use_bn:
            K.bn_fwd(ws, self.S, oc, st[self.bnA[bn_pass]], act.code, self.preA, self.hA, self.statsA[bn_pass], rows=rows,
                     leak=act.leak, u_ptr=self.S.ptr(r0), pre_ptr=self.preA.ptr(r0), h_ptr=self.hA.ptr(r0))
        else:
            _lib.call('tdg_bias_act', dt, self.S.ptr(r0), rows, oc, self.S.cs, None, act.code, act.leak, self.hA.ptr(r0), K.stream())
        self.convB.fwd(self.hA.ptr(r0), self.T.ptr(r0), rn, K.epilogue(bias=st[self.names['Bb']]))
        branch = self.T
        if sp.use_bn:
            K.bn_fwd(ws, self.T, oc, st[self.bnB[bn_pass]], K.ACT_NONE, self.preB, self.preB, self.statsB[bn_pass], rows=rows,
                     u_ptr=self.T.ptr(r0), pre_ptr=self.preB.ptr(r0), h_ptr=self.preB.ptr(r0))
            branch = self.preB
        K.add_act(dt, branch.ptr(r0), self.S.ptr(r0), nel, L.h.ptr(r0), act.code, act.leak)          # act(branch + shortcut)

    def backward(self, r0, rn, q0, qn, bn_pass, want_params, beta, need_in, mmode, mleak, mptr):
        st, g, sp, L, ws = self.seq.store, self.seq.store.grad, self.spec, self.L, self.seq.ws
        act, oc, dt = L.act, sp.out_size, self.seq.dtype
        hw = self.S.h * self.S.w
        rows, nel = rn * hw, rn * self.S.image_elems
        sink = self._sink
        # dZ = dOut * act'(out)
        _lib.call('tdg_act_bwd', dt, L.gout.ptr(r0), L.h.ptr(r0), nel, act.code, act.leak, self.dZ.ptr(r0), K.stream())
        if sp.use_bn:
            K.bn_bwd(ws, self.dZ, self.preB, oc, st[self.bnB[bn_pass]], self.statsB[bn_pass], K.ACT_NONE, self.dT,
                     g(self.bnB[bn_pass]) if want_params else sink, rows=rows, beta_acc=0.0,
                     dh_ptr=self.dZ.ptr(r0), pre_ptr=self.preB.ptr(r0), du_ptr=self.dT.ptr(r0))
            dT = self.dT
        else:
            dT = self.dZ
        if want_params and qn > 0:
            K.bias_grad(ws, dT, oc, g(self.names['Bb']), rows=qn * hw, beta=beta, dy_ptr=dT.ptr(q0))
            self.convB.bwd_filter(self.hA.ptr(q0), dT.ptr(q0), g(self.names['Bw']), qn, beta)
        if sp.use_bn:
            self.convB.bwd_data(dT.ptr(r0), self.dhA.ptr(r0), rn)
            K.bn_bwd(ws, self.dhA, self.preA, oc, st[self.bnA[bn_pass]], self.statsA[bn_pass], act.code, self.dS,
                     g(self.bnA[bn_pass]) if want_params else sink, rows=rows, leak=act.leak, beta_acc=0.0,
                     dh_ptr=self.dhA.ptr(r0), pre_ptr=self.preA.ptr(r0), du_ptr=self.dS.ptr(r0))
        else:
            m = K.MASK_LRELU if act.code == K.ACT_LRELU else (K.MASK_RELU if act.code == K.ACT_RELU else K.MASK_NONE)
            if m == K.MASK_NONE and act.code != K.ACT_NONE:
                raise NotImplementedError('residual %s: activation %s without batch norm' % (sp.name, act.name))
            self.convB.bwd_data(dT.ptr(r0), self.dS.ptr(r0), rn, K.epilogue(mask_mode=m, leak=act.leak, mask_src=self.hA.ptr(r0)))
        K.add_act(dt, self.dS.ptr(r0), self.dZ.ptr(r0), nel, self.dS.ptr(r0))                # + the shortcut's gradient
        if want_params and qn > 0:
            K.bias_grad(ws, self.dS, oc, g(self.names['Ab']), rows=qn * hw, beta=beta, dy_ptr=self.dS.ptr(q0))
            self.convA.bwd_filter(L.inp.ptr(q0), self.dS.ptr(q0), g(self.names['Aw']), qn, beta)
        if need_in:
            self.convA.bwd_data(self.dS.ptr(r0), L.gin.ptr(r0), rn, K.epilogue(mask_mode=mmode, leak=mleak, mask_src=mptr))


def _alias_rows(act, rpi, h, w, c):
    """View `act` ([n, ...]) as [n*rpi, h, w, c]."""
    if act is None:
        return None
    if rpi == 1:
        return _alias(act, h, w, c)
    if act.cs != act.c or K.pad_channels(c) != c:
        raise ValueError('row-splitting reshape needs unpadded layouts')
    return K.Act(act.n * rpi, h, w, c, act.dtype, act.buf.device, c, act.buf)


def _sub(t, start, count):
    return t[start:start + count]
