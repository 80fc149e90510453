"""ctypes binding of lib3dgan_hip.so (C ABI declared in include/tdg.h).

There is no CPU fallback: if the shared library is missing or a call fails, a RuntimeError
carrying the status code and `tdg_last_error()` is raised (error
convention).  PyTorch-ROCm tensors are used only as device-memory containers: every kernel
argument is a raw `data_ptr()` and the launch stream is torch's current HIP stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TDG_LIB_PATH') or os.path.join(_HERE, 'lib3dgan_hip.so')   # override: diagnostic builds only

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
MASK_NONE, MASK_LRELU, MASK_RELU = 0, 1, 2
COL_NONE, COL_SUM, COL_BN = 0, 1, 2


class ConvDesc(C.Structure):
    """TdgConvDesc (include/tdg.h)."""
    _fields_ = [(n, C.c_int32) for n in
                ('n', 'h', 'w', 'c', 'cs', 'oh', 'ow', 'k', 'ks', 'kh', 'kw', 'stride', 'pad_t', 'pad_l', 'dtype')]


class Epilogue(C.Structure):
    """TdgEpilogue (include/tdg.h)."""
    _fields_ = [('bias', C.c_void_p), ('act', C.c_int32), ('leak', C.c_float),
                ('mask_mode', C.c_int32), ('mask_src', C.c_void_p), ('accumulate', C.c_int32),
                ('col_partial', C.c_void_p), ('col_partial_bytes', C.c_size_t), ('col_mode', C.c_int32),
                ('col_images', C.c_int32), ('col_nblk_out', C.POINTER(C.c_int32)),
                ('splitk_ws', C.c_void_p), ('splitk_ws_bytes', C.c_size_t)]


class PackJob(C.Structure):
    """TdgPackJob (include/tdg.h)."""
    _fields_ = [('desc', ConvDesc), ('w', C.c_void_p), ('packed_fwd', C.c_void_p), ('packed_bwd', C.c_void_p)]


class LaunchRecord(C.Structure):
    """TdgLaunchRecord (include/tdg.h)."""
    _fields_ = [('kernel', C.c_char * 64), ('ms', C.c_double), ('flops', C.c_double)]


_vp, _i, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64
_PD, _PE = C.POINTER(ConvDesc), C.POINTER(Epilogue)

# name -> (restype, argtypes); every symbol include/tdg.h declares
SIGNATURES = {
    'tdg_last_error': (C.c_char_p, []),
    'tdg_version': (_i, []),
    'tdg_last_kernel': (C.c_char_p, []),
    'tdg_timing_begin': (_i, []),
    'tdg_timing_end': (_i, [C.POINTER(LaunchRecord), _i, C.POINTER(_i)]),
    'tdg_packed_filter_fwd_bytes': (_sz, [_PD]),
    'tdg_packed_filter_bwd_bytes': (_sz, [_PD]),
    'tdg_pack_filter_fwd': (_i, [_PD, _vp, _vp, _vp]),
    'tdg_pack_filter_bwd': (_i, [_PD, _vp, _vp, _vp]),
    'tdg_pack_filters': (_i, [C.POINTER(PackJob), _i, _vp]),
    'tdg_conv2d_fwd': (_i, [_PD, _i, _vp, _vp, _vp, _PE, _vp]),
    'tdg_conv2d_bwd_data': (_i, [_PD, _i, _vp, _vp, _vp, _PE, _vp]),
    'tdg_conv2d_bwd_filter_workspace_bytes': (_sz, [_PD, _i]),
    'tdg_conv2d_bwd_filter': (_i, [_PD, _i, _vp, _vp, _vp, _f, _vp, _sz, _vp]),
    'tdg_conv2d_bwd_filter2': (_i, [_PD, _i, _vp, _i, _vp, _vp, _vp, _f, _vp, _sz, _vp]),
    'tdg_rowdot': (_i, [_i, _vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    'tdg_rowouter': (_i, [_i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    'tdg_col_finalize_sum': (_i, [_vp, _i, _i, _vp, _f, _vp]),
    'tdg_instance_norm_fwd': (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _f, _vp, _i, _vp, _vp]),
    'tdg_instance_norm_bwd': (_i, [_i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _f, _vp, _sz, _vp]),
    'tdg_add_act': (_i, [_i, _vp, _vp, _sz, _i, _f, _vp, _vp]),
    'tdg_bn_fwd_from_partials': (_i, [_i, _vp, _i, _i, _i, _vp, _f, _i, _f, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    'tdg_colsum_weighted': (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _f, _vp, _sz, _vp]),
    'tdg_colsum_workspace_bytes': (_sz, [_i, _i]),
    'tdg_bn_workspace_bytes': (_sz, [_i, _i]),
    'tdg_bn_fwd': (_i, [_i, _vp, _i, _i, _i, _vp, _f, _i, _f, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    'tdg_bn_fwd_groups': (_i, [_i, _vp, _i, _i, _i, _i, _vp, _f, _i, _f, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    'tdg_bn_bwd': (_i, [_i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _f, _vp, _vp, _f, _vp, _f, _vp, _sz, _vp]),
    'tdg_affine_cast_pair': (_i, [_i, _vp, _i, _vp, _i, _sz, _f, _f, _vp, _vp, _vp]),
    'tdg_bias_act': (_i, [_i, _vp, _i, _i, _i, _vp, _i, _f, _vp, _vp]),
    'tdg_act_bwd': (_i, [_i, _vp, _vp, _sz, _i, _f, _vp, _vp]),
    'tdg_affine_cast': (_i, [_i, _vp, _sz, _f, _f, _vp, _vp]),
    'tdg_affine_cast_rows': (_i, [_i, _vp, _i, _i, _i, _f, _f, _vp, _vp]),
    'tdg_cast_to_f32': (_i, [_i, _vp, _sz, _vp, _vp]),
    'tdg_cast_from_f32': (_i, [_i, _vp, _sz, _vp, _vp]),
    'tdg_gp_interp': (_i, [_i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    'tdg_sumsq': (_i, [_i, _vp, _sz, _vp, _f, _vp, _sz, _vp]),
    'tdg_reduce_workspace_bytes': (_sz, [_sz]),
    'tdg_mean_f32': (_i, [_vp, _i, _vp, _vp]),
    'tdg_sum_f32': (_i, [_vp, _i, _vp, _f, _vp]),
    'tdg_mean_segments_f32': (_i, [_vp, _i, _i, _vp, _vp]),
    'tdg_gan_logloss': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    'tdg_p2p_xent': (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'tdg_p2p_l1': (_i, [_i, _vp, _vp, _i, _i, _f, _vp, _i, _vp, _vp, _sz, _vp]),
    'tdg_vae_reparam': (_i, [_i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp]),
    'tdg_vae_reparam_bwd': (_i, [_i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp]),
    'tdg_vae_reparam_bwd_kl': (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _f, _i, _i, _vp, _i, _vp]),
    'tdg_gp_rows': (_i, [_i, _vp, _i, _i, _f, _vp, _vp, _vp]),
    'tdg_vae_kl': (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'tdg_vae_bce': (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'tdg_dropout': (_i, [_i, _vp, _i, _i, _i, _vp, _f, _vp]),
    'tdg_l1_loss': (_i, [_i, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    'tdg_gp_scalars': (_i, [_vp, _f, _vp, _vp]),
    'tdg_gp_sumsq': (_i, [_i, _vp, _sz, _vp, _f, _vp, _vp, _sz, _vp]),
    'tdg_scale_by_dev': (_i, [_i, _vp, _sz, _vp, _vp, _vp]),
    'tdg_fill_f32': (_i, [_vp, _sz, _f, _vp]),
    'tdg_bias_grad': (_i, [_i, _vp, _i, _i, _i, _vp, _f, _vp, _sz, _vp]),
    'tdg_adam_step': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _vp]),
    'tdg_adam_step_dev': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _vp, _vp]),
    'tdg_add_i32': (_i, [_vp, _i, _vp]),
    'tdg_rmsprop_step': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _vp]),
    'tdg_rmsprop_centered_step': (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _vp]),
    'tdg_adagrad_step': (_i, [_vp, _vp, _vp, _sz, _f, _f, _vp]),
    'tdg_adadelta_step': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _vp]),
    'tdg_ftrl_step': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _vp]),
    'tdg_sgd_momentum_step': (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _vp]),
    'tdg_clamp': (_i, [_vp, _sz, _f, _f, _vp]),
    'tdg_check_finite': (_i, [_vp, _sz, _vp, _vp]),
    'tdg_random_normal': (_i, [_i, _u64, _u64, _u64, _sz, _vp, _vp]),
    'tdg_random_uniform_f32': (_i, [_u64, _u64, _u64, _sz, _vp, _vp]),
    'tdg_random_normal_dev': (_i, [_i, _u64, _u64, _vp, _sz, _vp, _vp]),
    'tdg_random_uniform_f32_dev': (_i, [_u64, _u64, _vp, _sz, _vp, _vp]),
    'tdg_png_unfilter': (_i, [C.c_char_p, _i, _i, _i, _vp]),
    'tdg_jpeg_info': (_i, [C.c_char_p, _sz, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'tdg_jpeg_decode': (_i, [C.c_char_p, _sz, _vp, _sz]),
    'tdg_shuffle_draw': (_i, [_vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int64, _vp]),
    'tdg_gather_rows': (_i, [_vp, C.c_int64, _sz, _vp, C.c_int64, _vp]),
}

_lib = None


class TdgError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and declare every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TdgError('%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       '(3dgan_amd/csrc/build.sh). There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().tdg_last_error()
        raise TdgError('%s failed with status %d: %s' % (what, rc, msg.decode() if msg else ''))


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    check(getattr(load(), name)(*args), name)
