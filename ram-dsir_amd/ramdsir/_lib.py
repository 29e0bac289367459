"""ctypes binding of libramdsir_hip.so (include/ramdsir.h).  No fallback: if the HIP library is
missing or a call fails, this raises -- the product path never routes around the kernels."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAMDSIR_DEBUG_LIB=1: the debug build (`make debug`), whose dispatch honours RD_* environment overrides (csrc/common.h
# rd_switch); the product library has none
LIB_PATH = os.path.join(_HERE, 'libramdsir_hip_dbg.so' if os.environ.get('RAMDSIR_DEBUG_LIB') == '1' else 'libramdsir_hip.so')

RD_F32, RD_BF16 = 0, 1
MAXG = 16
STAT_SLOTS = 64
STAT_SLOTS_FOLD = 8
FIN_OWNER = 1
FIN_MAX_G = 8          # groups a folded finalize may have (csrc/bn_fin.h)
SRC_RAW, SRC_AFF, SRC_AFFACT, SRC_POOL, SRC_UP, SRC_BNBWD = range(6)
DST_PLAIN, DST_POOL, DST_UPY, DST_NONE = range(4)

vp, fp, i32, i64, f32 = C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float


class RdSrc(C.Structure):
    _fields_ = [('ptr', vp), ('ptr2', vp), ('scale', fp), ('shift', fp), ('q', fp), ('out', vp), ('mode', i32), ('C', i32),
                ('slope', f32), ('n_off', i32), ('g_fixed', i32), ('fin_flags', i32), ('fin', vp)]


class RdDst(C.Structure):
    _fields_ = [('g', vp), ('z', vp), ('scale', fp), ('shift', fp), ('bstats', fp), ('kind', i32), ('act', i32),
                ('accumulate', i32), ('Cd', i32), ('slope', f32), ('n_off', i32), ('g_fixed', i32), ('pad_', i32)]


class RdConv(C.Structure):
    _fields_ = [('src', RdSrc * 2), ('nsrc', i32), ('taps', i32), ('w', vp), ('bias', fp), ('CinPad', i32),
                ('CoutPad', i32), ('N', i32), ('H', i32), ('W', i32), ('Cin', i32), ('Cout', i32), ('G', i32),
                ('gstart', i32 * (MAXG + 1)), ('emode', i32), ('out', vp), ('stats', fp), ('dst', RdDst * 2),
                ('c_split', i32), ('cu_limit', i32), ('w_tap_rows', i32), ('stat_slots', i32)]


class RdWgrad(C.Structure):
    _fields_ = [('a', RdSrc * 2), ('na', i32), ('taps', i32), ('dz', RdSrc), ('N', i32), ('H', i32), ('W', i32),
                ('Cin', i32), ('Cout', i32), ('G', i32), ('gstart', i32 * (MAXG + 1)), ('partial', fp), ('dW', fp),
                ('beta', f32), ('cu_limit', i32)]


class RdBnFwd(C.Structure):
    _fields_ = [('stats', fp), ('conv_bias', fp), ('scale', fp), ('shift', fp), ('mean', fp), ('invstd', fp), ('gamma', fp * MAXG),
                ('beta', fp * MAXG), ('running_mean', fp * MAXG), ('running_var', fp * MAXG),
                ('num_batches_tracked', vp * MAXG), ('count', f32 * MAXG), ('C', i32), ('G', i32), ('eps', f32),
                ('momentum', f32), ('training', i32), ('nslots', i32)]


class RdBnBwd(C.Structure):
    _fields_ = [('bstats', fp), ('mean', fp), ('invstd', fp), ('gamma', fp * MAXG), ('dgamma', fp * MAXG),
                ('dbeta', fp * MAXG), ('P', fp), ('Q', fp), ('R', fp), ('count', f32 * MAXG), ('C', i32), ('G', i32),
                ('fstats', fp), ('conv_bias', fp), ('dbias', fp), ('nslots', i32), ('pad_', i32)]


class RdSegLoss(C.Structure):
    _fields_ = [('logits', vp), ('target', vp), ('dlogits', vp), ('partial', fp), ('losses_out', fp), ('B', i32),
                ('H', i32), ('W', i32), ('K', i32), ('kind', i32), ('consistency', i32), ('cons_weight', f32),
                ('dlogits_cstride', i32)]


class RdAdam(C.Structure):
    _fields_ = [('param', fp), ('grad', fp), ('exp_avg', fp), ('exp_avg_sq', fp), ('n', i64), ('n_half_lr', i64),
                ('iter', vp), ('hyper_out', fp), ('base_lr', f32), ('total_iters', i32), ('beta1', f32),
                ('beta2', f32), ('eps', f32), ('pad_', i32)]


class RdRam(C.Structure):
    _fields_ = [('src', fp), ('trg', fp), ('lam', fp), ('out_img', vp), ('out_freq', vp), ('workspace', vp),
                ('tw_w', fp), ('tw_h', fp), ('B', i32), ('H', i32), ('W', i32), ('C', i32), ('b', i32),
                ('clip_lo', f32), ('clip_hi', f32), ('scale', f32), ('offset', f32), ('out_cstride', i32), ('src_u8', i32),
                ('div', f32), ('pad_', i32), ('trg_amp', fp), ('dft_tables', vp)]


class RdPackEntry(C.Structure):
    _fields_ = [('src_off', i64), ('dst_off', i64), ('start', i64), ('Cout', i32), ('Cin', i32), ('taps', i32),
                ('transpose', i32), ('RowPad', i32), ('ColPad', i32)]


class RdLaunch(C.Structure):
    """rd_launch_t (include/ramdsir.h): one entry of a native launch list."""
    _fields_ = [('op', i32), ('lane', i32), ('wait_main', i32), ('nargs', i32), ('a', C.c_uint64 * 18)]


# RD_OP_* (include/ramdsir.h) by entry-point name
OP_FORK, OP_JOIN = 1, 2
OP_CODES = {name: 10 + i for i, name in enumerate((
    'rd_conv', 'rd_wgrad', 'rd_conv_bwd_fused', 'rd_conv_bwd_fused_reduce', 'rd_pack_weights_batched',
    'rd_bn_finalize_fwd', 'rd_bn_finalize_bwd', 'rd_gn_finalize_fwd', 'rd_gn_finalize_bwd',
    'rd_up_stats', 'rd_bn_stats', 'rd_up_bwd', 'rd_pool_fwd', 'rd_pool_bwd', 'rd_bn_apply',
    'rd_nchw_to_nhwc', 'rd_nhwc_to_nchw', 'rd_grad_in', 'rd_colsum',
    'rd_seg_loss', 'rd_rec_loss', 'rd_adam_step', 'rd_zero', 'rd_ram_mix'))}

_SIGS = {
    'rd_ram_workspace': (i64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'rd_ram_mix': (C.c_int, [C.POINTER(RdRam), C.c_int, vp]),
    'rd_ram_dft_tables_bytes': (i64, [C.c_int, C.c_int, C.c_int]),
    'rd_ram_dft_tables': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    'rd_ram_amp_workspace': (i64, [C.c_int, C.c_int, C.c_int]),
    'rd_ram_amp': (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, vp, fp, fp, vp]),
    'rd_ram_mutate': (C.c_int, [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp]),
    'rd_pack_weights_batched': (C.c_int, [fp, vp, vp, C.c_int, i64, C.c_int, vp]),
    'rd_conv': (C.c_int, [C.POINTER(RdConv), C.c_int, vp]),
    'rd_wgrad_workspace': (i64, [C.POINTER(RdWgrad), C.c_int]),
    'rd_wgrad': (C.c_int, [C.POINTER(RdWgrad), C.c_int, vp]),
    'rd_conv_honours_src_out': (C.c_int, [C.POINTER(RdConv), C.c_int]),
    'rd_conv_bwd_fused_ok': (C.c_int, [C.POINTER(RdConv), C.POINTER(RdWgrad), C.c_int]),
    'rd_conv_bwd_fused_workspace': (i64, [C.POINTER(RdConv), C.POINTER(RdWgrad), C.c_int]),
    'rd_conv_bwd_fused': (C.c_int, [C.POINTER(RdConv), C.POINTER(RdWgrad), C.c_int, vp]),
    'rd_conv_bwd_fused_reduce': (C.c_int, [C.POINTER(RdConv), C.POINTER(RdWgrad), C.c_int, vp]),
    'rd_pack_weights': (C.c_int, [fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'rd_packed_elems': (i64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'rd_bn_finalize_fwd': (C.c_int, [C.POINTER(RdBnFwd), vp]),
    'rd_bn_finalize_bwd': (C.c_int, [C.POINTER(RdBnBwd), vp]),
    'rd_gn_finalize_fwd': (C.c_int, [C.POINTER(RdBnFwd), vp]),
    'rd_gn_finalize_bwd': (C.c_int, [C.POINTER(RdBnBwd), vp]),
    'rd_bn_apply': (C.c_int, [vp, vp, vp, fp, fp, fp, f32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(i32), C.c_int, vp]),
    'rd_up_stats': (C.c_int, [vp, fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(i32), C.c_int, C.c_int, vp]),
    'rd_bn_stats': (C.c_int, [vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(i32), C.c_int, vp]),
    'rd_pool_fwd': (C.c_int, [vp, fp, fp, f32, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(i32), C.c_int, vp, C.c_int, vp]),
    'rd_pool_bwd': (C.c_int, [vp, vp, fp, fp, f32, C.c_int, vp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(i32), C.c_int, C.c_int, vp]),
    'rd_up_bwd': (C.c_int, [vp, vp, vp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(i32),
                            C.c_int, vp, C.c_int, vp]),
    'rd_nchw_to_nhwc': (C.c_int, [fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'rd_nhwc_to_nchw': (C.c_int, [vp, fp, fp, fp, C.c_int, f32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(i32), C.c_int, vp]),
    'rd_grad_in': (C.c_int, [fp, vp, vp, fp, fp, fp, C.c_int, f32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.POINTER(i32), C.c_int, vp]),
    'rd_colsum': (C.c_int, [vp, fp, fp, i64, C.c_int, C.c_int, f32, C.c_int, vp]),
    'rd_seg_loss_workspace': (i64, [C.POINTER(RdSegLoss)]),
    'rd_seg_loss': (C.c_int, [C.POINTER(RdSegLoss), C.c_int, vp]),
    'rd_rec_loss': (C.c_int, [vp, vp, vp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(i32), f32,
                              C.c_int, vp]),
    'rd_rec_loss_workspace': (i64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'rd_adam_step': (C.c_int, [C.POINTER(RdAdam), vp]),
    'rd_zero': (C.c_int, [C.POINTER(vp), C.POINTER(i64), C.c_int, vp]),
    'rd_run_list': (C.c_int, [C.POINTER(RdLaunch), C.c_int, C.POINTER(vp), C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    'rd_join_lanes': (C.c_int, [C.POINTER(vp), C.c_int, C.c_uint32]),
    'rd_run_list_threads': (C.c_int, [C.c_int]),
    'rd_run_list_bind_fork_events': (C.c_int, [C.c_int]),
    'rd_run_list_fork_counts': (None, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    'rd_run_list_fork_plan': (C.c_int, [C.POINTER(RdLaunch), C.c_int, C.POINTER(C.c_ubyte)]),
    'rd_box_probe': (C.c_int, [C.c_int, vp, vp, i64, vp]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def lib():
    """The loaded library (cached).  Raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libramdsir_hip.so is not built: run `make -C ram-dsir_amd/csrc` '
                               '(or __graft_entry__.build()); there is no CPU fallback')
        # ONE HIP runtime per process: the library's libamdhip64.so.7 must be the copy PyTorch ships and has initialised (the streams and
        # device pointers handed to the entry points are that runtime's).  Loaded before torch, the dynamic linker would bind it to
        # /opt/rocm/lib's copy and the first launch on a torch stream fails with hipErrorNoDevice (__graft_entry__.build() followed by
        # smoke() in one process did) -- so torch comes first, whatever the caller's import order.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(err, what=''):
    if err != 0:
        raise RuntimeError('ramdsir HIP call failed (%s): error %d' % (what, err))


def pack_arg(value, ctype):
    """One argument of an entry point as the 64-bit word rd_launch_t.a[] carries (include/ramdsir.h): pointers and integers as they
    are, a float as its bit pattern."""
    import struct
    if value is None:
        return 0
    if ctype is f32 or ctype is C.c_float:
        return struct.unpack('<I', struct.pack('<f', float(value)))[0]
    if ctype in (C.c_int, i32, i64, C.c_int64, C.c_uint32):
        return int(value) & 0xFFFFFFFFFFFFFFFF
    # pointer-like: byref(obj), a ctypes array / structure, a c_void_p, or a raw address
    if hasattr(value, '_obj'):
        return C.addressof(value._obj)
    if isinstance(value, (C.Array, C.Structure)):
        return C.addressof(value)
    if isinstance(value, C.c_void_p):
        return value.value or 0
    return int(value)


def gstart_array(gs):
    a = (i32 * (MAXG + 1))()
    for i, v in enumerate(gs):
        a[i] = int(v)
    return a


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
