"""Thin typed wrappers over the C ABI (include/tdg.h): torch tensors in, raw pointers out.

Nothing here computes: each function marshals pointers/sizes and enqueues one library call
on torch's current HIP stream.  `Act` describes an NHWC activation buffer with an explicit
channel stride (padding channels are kept at zero; see DESIGN.md "Data layout in HBM").
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (F32, BF16, ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID,  # noqa: F401
                   MASK_NONE, MASK_LRELU, MASK_RELU, COL_NONE, COL_SUM, COL_BN, ConvDesc, Epilogue)

TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16}
ELEM_SIZE = {F32: 4, BF16: 2}


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, byte_offset=0):
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr() + byte_offset)


def pad_channels(c, multiple=8):
    """Channel stride of an activation buffer: a multiple of 8 keeps the 16-byte vector gather
    legal in both dtypes.  Images (c = 1, 3, 4) are padded to 8 channels too: the extra MFMA work
    on zero channels is far cheaper than the element-wise gather a compact layout would need."""
    return (c + multiple - 1) // multiple * multiple


class Act:
    """[n, h, w, c] activation in HBM with channel stride cs (zero-filled at allocation)."""

    def __init__(self, n, h, w, c, dtype, device, cs=None, buf=None):
        self.n, self.h, self.w, self.c = n, h, w, c
        self.cs = pad_channels(c) if cs is None else cs
        self.dtype = dtype
        self.buf = buf if buf is not None else torch.zeros(n * h * w * self.cs, dtype=TORCH_DTYPE[dtype], device=device)

    @property
    def rows(self):
        return self.n * self.h * self.w

    @property
    def image_elems(self):
        return self.h * self.w * self.cs

    def ptr(self, image=0):
        return ptr(self.buf, image * self.image_elems * ELEM_SIZE[self.dtype])

    def view(self, image0, count):
        """Sub-batch [image0, image0+count) sharing storage."""
        e = self.image_elems
        return Act(count, self.h, self.w, self.c, self.dtype, self.buf.device, self.cs,
                   self.buf[image0 * e:(image0 + count) * e])

    def like(self):
        return Act(self.n, self.h, self.w, self.c, self.dtype, self.buf.device, self.cs)

    def window(self, c_off, c):
        """Channels [c_off, c_off + c) of every pixel as an Act sharing storage (zero-copy concat:
        producers write their half of a skip-concatenated tensor in place)."""
        assert c_off + c <= self.cs
        return Act(self.n, self.h, self.w, c, self.dtype, self.buf.device, self.cs, self.buf[c_off:])

    # host <-> device helpers (tests / checkpoints only)
    def set(self, array):
        t = torch.as_tensor(array, dtype=torch.float32).reshape(self.n, self.h, self.w, self.c)
        full = torch.zeros(self.n, self.h, self.w, self.cs, dtype=torch.float32)
        full[..., :self.c] = t
        self.buf.copy_(full.reshape(-1).to(self.buf.device, TORCH_DTYPE[self.dtype]))
        return self

    def get(self):
        return self.buf.float().reshape(self.n, self.h, self.w, self.cs)[..., :self.c].cpu().numpy()


def conv_desc(big, small, kh, kw, stride, pad_t, pad_l):
    """Descriptor for the conv between `big` (conv input / deconv output) and `small`."""
    assert big.dtype == small.dtype and big.n == small.n
    return ConvDesc(big.n, big.h, big.w, big.c, big.cs, small.h, small.w, small.c, small.cs,
                    kh, kw, stride, pad_t, pad_l, big.dtype)


_SPLITK = {}


def splitk_workspace(device):
    """One scratch buffer per device for split-K partial tiles (TdgEpilogue.splitk_ws): 64 MB covers the widest small-M
    layer (4096 rows x 1024 columns x 4 splits)."""
    buf = _SPLITK.get(device)
    if buf is None:
        buf = _SPLITK[device] = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    return buf


def epilogue(bias=None, act=ACT_NONE, leak=0.2, mask_mode=MASK_NONE, mask_src=None, accumulate=False):
    e = Epilogue()
    e.accumulate = 1 if accumulate else 0
    e.bias = bias.data_ptr() if bias is not None else None
    e.act, e.leak, e.mask_mode = act, leak, mask_mode
    e.mask_src = mask_src if isinstance(mask_src, int) or mask_src is None else mask_src.value
    return e


def colsum_epilogue(ws, rows, c, mode, images=0, **kw):
    """An epilogue that also asks the GEMM for per-row-tile column partials of the tile it stores (TdgEpilogue.col_partial):
    COL_SUM = the bias gradient of the tensor's layer, COL_BN = its batch statistics.  After the launch `nblk(e)` tells how
    many row tiles were written (0: the kernel variant chosen for this launch cannot -- run the separate reduction pass)."""
    e = epilogue(**kw)
    nblk_max = rows // 64 + 8                       # row tiles are >= 64 rows; <= 4 parity classes round up separately
    nbytes = nblk_max * 2 * c * 4
    buf = ws.ensure(nbytes)
    e.col_partial, e.col_partial_bytes, e.col_mode, e.col_images = buf.data_ptr(), nbytes, mode, images
    e._nblk = C.c_int32(0)
    e.col_nblk_out = C.pointer(e._nblk)
    return e


def nblk(e):
    return int(e._nblk.value) if hasattr(e, '_nblk') else 0


def bias_grad_from_partials(e, c, db, beta=0.0):
    _lib.call('tdg_col_finalize_sum', C.c_void_p(e.col_partial), nblk(e), c, ptr(db), beta, stream())


def bn_fwd_from_partials(e, u, c, beta, act, pre, h, stats, bias, rows=None, leak=0.2, eps=1e-3, u_ptr=None, pre_ptr=None, h_ptr=None):
    """pre=None: only h is written (a forward pass that no backward pass will follow)."""
    rows = u.rows if rows is None else rows
    _lib.call('tdg_bn_fwd_from_partials', u.dtype, u_ptr or u.ptr(), rows, c, u.cs, ptr(beta), eps, act, leak,
              (pre_ptr or pre.ptr()) if pre is not None else None, h_ptr or h.ptr(), h.cs, ptr(stats), C.c_void_p(e.col_partial),
              nblk(e), ptr(bias), stream())


class GemmTimer:
    """Optional HIP-event timing of the conv GEMM launches (bench.py roofline): events are
    recorded on torch's current stream, which is the stream every kernel is launched on."""

    def __init__(self):
        self.records = []          # (kind, start_event, end_event, flops)

    def wrap(self, kind, flops, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        symbol = _lib.load().tdg_last_kernel().decode()          # the kernel variant the library dispatched to
        self.records.append((kind, a, b, flops, symbol))

    def summary(self, by_symbol=False):
        """{kind or kernel symbol: (launches, total_ms, total_flops)} -- call after a synchronize.
        (bwd_filter timings include the slab_reduce launch that follows the GEMM.)"""
        out = {}
        for kind, a, b, fl, symbol in self.records:
            key = symbol if by_symbol else kind
            n, ms, f = out.get(key, (0, 0.0, 0.0))
            out[key] = (n + 1, ms + a.elapsed_time(b), f + fl)
        return out


TIMER = None       # set to a GemmTimer to time conv GEMM launches


def timing_begin():
    """Start the library's per-launch HIP-event timing of conv GEMM kernels (diagnostics; not during graph capture)."""
    _lib.call('tdg_timing_begin')


def timing_end(capacity=65536):
    """Stop it; returns [(kernel, ms, flops)] with one entry per kernel launch since timing_begin()."""
    buf = (_lib.LaunchRecord * capacity)()
    n = C.c_int(0)
    _lib.call('tdg_timing_end', buf, capacity, C.byref(n))
    return [(buf[i].kernel.decode(), buf[i].ms, buf[i].flops) for i in range(min(n.value, capacity))]


def pack_all(jobs):
    """All filters of a network in one library call (tdg_pack_filters); `jobs` is a ctypes array of PackJob."""
    _lib.call('tdg_pack_filters', jobs, len(jobs), stream())


def make_pack_jobs(job_list):
    return (_lib.PackJob * len(job_list))(*job_list)


class Conv:
    """One strided conv of the model: owns the packed filter operands and the split-K
    workspace; exposes the three GEMM forms of include/tdg.h."""

    def __init__(self, big, small, kh, kw, stride, pad_t, pad_l):
        self.big, self.small = big, small
        self.desc = conv_desc(big, small, kh, kw, stride, pad_t, pad_l)
        lib = _lib.load()
        dev = big.buf.device
        self.fwd_bytes = lib.tdg_packed_filter_fwd_bytes(C.byref(self.desc))
        self.bwd_bytes = lib.tdg_packed_filter_bwd_bytes(C.byref(self.desc))
        if self.fwd_bytes == 0 or self.bwd_bytes == 0:
            raise _lib.TdgError('conv descriptor rejected: %s' % lib.tdg_last_error().decode())
        self.w_fwd = torch.zeros(self.fwd_bytes, dtype=torch.uint8, device=dev)
        self.w_bwd = torch.zeros(self.bwd_bytes, dtype=torch.uint8, device=dev)
        self.ws_bytes = lib.tdg_conv2d_bwd_filter_workspace_bytes(C.byref(self.desc), big.n)
        self.ws = None
        self.filter_shape = (kh, kw, big.c, small.c)

    def workspace(self):
        if self.ws is None:
            self.ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=self.big.buf.device)
        return self.ws

    def pack(self, w, fwd=True, bwd=True):
        """w: f32 master filter [kh, kw, big.c, small.c] (device, contiguous)."""
        if fwd:
            _lib.call('tdg_pack_filter_fwd', C.byref(self.desc), ptr(w), ptr(self.w_fwd), stream())
        if bwd:
            _lib.call('tdg_pack_filter_bwd', C.byref(self.desc), ptr(w), ptr(self.w_bwd), stream())

    def pack_job(self, w, fwd=True, bwd=True):
        """The TdgPackJob equivalent of pack(w, fwd, bwd), for pack_all()."""
        return _lib.PackJob(self.desc, ptr(w), ptr(self.w_fwd) if fwd else None, ptr(self.w_bwd) if bwd else None)

    def flops(self, n_images):
        """Algorithmic FLOPs of any of the three GEMM forms on n_images (2 x MACs, padding taps counted)."""
        d = self.desc
        return 2.0 * n_images * d.oh * d.ow * d.kh * d.kw * d.c * d.k

    def _tag(self, form):
        d = self.desc
        return '%s/%s/%dx%d_c%d_k%d' % (form, 'bf16' if d.dtype == BF16 else 'f32', d.kh, d.kw, d.c, d.k)

    def _with_splitk(self, epi):
        """Every forward-type launch offers the library the device's split-K scratch (it decides per launch)."""
        if epi is None:
            epi = epilogue()
        ws = splitk_workspace(self.big.buf.device)
        epi.splitk_ws, epi.splitk_ws_bytes = ws.data_ptr(), ws.numel()
        return epi

    def fwd(self, x_ptr, y_ptr, n_images, epi=None):
        epi = self._with_splitk(epi)

        def go():
            _lib.call('tdg_conv2d_fwd', C.byref(self.desc), n_images, x_ptr, ptr(self.w_fwd), y_ptr,
                      C.byref(epi) if epi is not None else None, stream())
        TIMER.wrap(self._tag('fwd'), self.flops(n_images), go) if TIMER is not None else go()

    def bwd_data(self, y_ptr, x_ptr, n_images, epi=None):
        epi = self._with_splitk(epi)

        def go():
            _lib.call('tdg_conv2d_bwd_data', C.byref(self.desc), n_images, y_ptr, ptr(self.w_bwd), x_ptr,
                      C.byref(epi) if epi is not None else None, stream())
        TIMER.wrap(self._tag('bwd_data'), self.flops(n_images), go) if TIMER is not None else go()

    def bwd_filter(self, x_ptr, y_ptr, dw, n_images, beta=0.0):
        ws = self.workspace()

        def go():
            _lib.call('tdg_conv2d_bwd_filter', C.byref(self.desc), n_images, x_ptr, y_ptr, ptr(dw), beta,
                      ptr(ws), ws.numel(), stream())
        TIMER.wrap(self._tag('bwd_filter'), self.flops(n_images), go) if TIMER is not None else go()


    def bwd_filter2(self, x_ptr, n_first, x2_ptr, y_ptr, dw, n_images, beta=0.0):
        """Filter gradient over n_images whose big-side rows come from two tensors (tdg_conv2d_bwd_filter2)."""
        ws = self.workspace()

        def go():
            _lib.call('tdg_conv2d_bwd_filter2', C.byref(self.desc), n_images, x_ptr, n_first, x2_ptr, y_ptr, ptr(dw), beta,
                      ptr(ws), ws.numel(), stream())
        TIMER.wrap(self._tag('bwd_filter'), self.flops(n_images), go) if TIMER is not None else go()


class Workspace:
    """Scratch for the two-stage reductions (BN, bias grads, sumsq)."""

    def __init__(self, device, nbytes=8 << 20):
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._retired = []

    def ensure(self, nbytes):
        """Grow on demand.  A buffer that has been handed out may be baked into a captured hipGraph (every reduction
        writes and re-reads its partials through the pointer it was launched with), so an outgrown buffer is retired,
        never freed: replays of an earlier capture keep a valid block of their own."""
        if self.buf.numel() < nbytes:
            self._retired.append(self.buf)
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.buf.device)
        return self.buf


def bn_fwd(ws, u, c, beta, act, pre, h, stats, rows=None, leak=0.2, eps=1e-3, u_ptr=None, pre_ptr=None, h_ptr=None):
    rows = u.rows if rows is None else rows
    lib = _lib.load()
    need = lib.tdg_bn_workspace_bytes(rows, c)
    w = ws.ensure(need)
    _lib.call('tdg_bn_fwd', u.dtype, u_ptr or u.ptr(), rows, c, u.cs, ptr(beta), eps, act, leak,
              pre_ptr or pre.ptr(), h_ptr or h.ptr(), h.cs, ptr(stats), ptr(w), w.numel(), stream())


def bn_fwd_groups(ws, u, c, beta, act, pre, h, stats, rows_per_group, ngroups, leak=0.2, eps=1e-3, u_ptr=None, pre_ptr=None, h_ptr=None):
    """bn_fwd over ngroups batches behind each other, each with its own statistics (stats: [ngroups][2][c]); pre=None: only h."""
    need = ngroups * _lib.load().tdg_bn_workspace_bytes(rows_per_group, c)
    w = ws.ensure(need)
    _lib.call('tdg_bn_fwd_groups', u.dtype, u_ptr or u.ptr(), rows_per_group, ngroups, c, u.cs, ptr(beta), eps, act, leak,
              (pre_ptr or pre.ptr()) if pre is not None else None, h_ptr or h.ptr(), h.cs, ptr(stats), ptr(w), w.numel(), stream())


def bn_bwd(ws, dh, pre, c, beta, stats, act, du, dbeta, rows=None, leak=0.2, beta_acc=0.0,
           dh_ptr=None, pre_ptr=None, du_ptr=None, dbias=None, dbias_acc=0.0):
    """dbias: also the bias gradient of the conv in front of the batch norm (column sums of du), from the same pass."""
    rows = dh.rows if rows is None else rows
    lib = _lib.load()
    need = lib.tdg_bn_workspace_bytes(rows, c)
    w = ws.ensure(need)
    _lib.call('tdg_bn_bwd', dh.dtype, dh_ptr or dh.ptr(), dh.cs, pre_ptr or pre.ptr(), rows, c, pre.cs, ptr(beta), ptr(stats),
              act, leak, du_ptr or du.ptr(), ptr(dbeta), beta_acc, ptr(dbias) if dbias is not None else None, dbias_acc,
              ptr(w), w.numel(), stream())


def bias_grad(ws, dy, c, db, rows=None, beta=0.0, dy_ptr=None):
    rows = dy.rows if rows is None else rows
    need = _lib.load().tdg_bn_workspace_bytes(rows, c)
    w = ws.ensure(need)
    _lib.call('tdg_bias_grad', dy.dtype, dy_ptr or dy.ptr(), rows, c, dy.cs, ptr(db), beta, ptr(w), w.numel(), stream())


def colsum_weighted(ws, dtype, x_ptr, rows, cols, cs, coef, dw, beta=0.0):
    need = _lib.load().tdg_colsum_workspace_bytes(rows, cols)
    w = ws.ensure(need)
    _lib.call('tdg_colsum_weighted', dtype, x_ptr, rows, cols, cs, ptr(coef), ptr(dw), beta, ptr(w), w.numel(), stream())


def sumsq(ws, dtype, x_ptr, n, acc, beta=0.0):
    w = ws.ensure(4096)
    _lib.call('tdg_sumsq', dtype, x_ptr, n, ptr(acc), beta, ptr(w), w.numel(), stream())


def in_fwd(u, n, c, scale, shift, act, h, stats, leak=0.2, eps=1e-3, u_ptr=None, h_ptr=None):
    """Instance norm + activation of n images of `u` into `h` (hem/ops/images.py:73-89)."""
    _lib.call('tdg_instance_norm_fwd', u.dtype, u_ptr or u.ptr(), n, u.h * u.w, c, u.cs, ptr(scale), ptr(shift), eps, act, leak,
              h_ptr or h.ptr(), h.cs, ptr(stats), stream())


def in_bwd(ws, dh, u, n, c, scale, shift, stats, act, du, dscale, dshift, leak=0.2, beta=0.0, dh_ptr=None, u_ptr=None, du_ptr=None):
    w = ws.ensure(n * 2 * c * 4)
    _lib.call('tdg_instance_norm_bwd', dh.dtype, dh_ptr or dh.ptr(), dh.cs, u_ptr or u.ptr(), n, u.h * u.w, c, u.cs, ptr(scale),
              ptr(shift), ptr(stats), act, leak, du_ptr or du.ptr(), ptr(dscale), ptr(dshift), beta, ptr(w), w.numel(), stream())


def add_act(dtype, a_ptr, b_ptr, n_elems, out_ptr, act=ACT_NONE, leak=0.2):
    """out = act(a + b) over n_elems elements of one layout."""
    _lib.call('tdg_add_act', dtype, a_ptr, b_ptr, n_elems, act, leak, out_ptr, stream())
