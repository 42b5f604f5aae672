"""ctypes binding of libxps.so (include/xps.h).  The product path has NO fallback: if
the library is missing or a call fails, an exception is raised."""
import ctypes as C
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('XPS_LIB_OVERRIDE') or os.path.join(_PKG, 'libxps.so')   # override: A/B builds in tools/
HEADER_PATH = os.path.join(_PKG, '..', 'include', 'xps.h')

_lib = None


class RowMap(C.Structure):
    """xps_rowmap: row i lives at (i // rpg) * gs + (i % rpg) * ld."""
    _fields_ = [('gs', C.c_int64), ('ld', C.c_int64), ('rpg', C.c_int32), ('fmt', C.c_int32)]


_rowmaps = {}


def rowmap(ld, rpg=1 << 30, gs=0, fmt=0):
    """xps_rowmap descriptor.  Instances are cached and shared (the library only reads them; embedding one in an
    xps_tn_problem copies it): building a ctypes struct costs ~2 us and a training step needs ~40 of them.
    fmt = 1 (XPS_FMT_SPLIT4): the GEMM input operand holds pre-split bf16 hi / lo groups (include/xps.h)."""
    key = (ld, rpg, gs, fmt)
    r = _rowmaps.get(key)
    if r is None:
        r = RowMap(int(gs), int(ld), int(rpg), int(fmt))
        if len(_rowmaps) < 4096:
            _rowmaps[key] = r
    return r


class TnProblem(C.Structure):
    """xps_tn_problem"""
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p), ('colsum_a', C.c_void_p),
                ('ra', RowMap), ('rb', RowMap), ('rc', RowMap),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('accumulate', C.c_int32)]


class XpsError(RuntimeError):
    pass


_vp, _i, _i64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
_rm = C.POINTER(RowMap)

# name -> (restype, argtypes); must list every function declared in include/xps.h
SIGNATURES = {
    'xps_last_error': (C.c_char_p, []),
    'xps_abi_version': (_i, []),
    'xps_stream_create_low_priority': (_i, [_vp]),
    'xps_stream_destroy': (_i, [_vp]),
    'xps_set_gemm_precision': (_i, [_i]),
    'xps_get_gemm_precision': (_i, []),
    'xps_set_gemm_big_tiles': (_i, [_i]),
    'xps_get_gemm_big_tiles': (_i, []),
    'xps_gemm_nt_f32': (_i, [_vp, _rm, _vp, _rm, _vp, _rm, _vp, _i, _i, _i, _i, _vp]),
    'xps_gemm_nn_f32': (_i, [_vp, _rm, _vp, _rm, _vp, _rm, _i, _i, _i, _i, _vp]),
    'xps_gemm_nt_multi_f32': (_i, [_vp, _rm, _vp, _rm, _vp, _rm, _vp, _i, _i, _i, _i, _vp]),
    'xps_gemm_nn2_f32': (_i, [_vp, _vp, _i, _vp, _vp, _i, _rm, _rm, _vp, _rm, _i, _i, _i, _vp]),
    'xps_gemm_tn_f32_workspace': (_sz, [_i, _i, _i]),
    'xps_gemm_tn_f32': (_i, [_vp, _rm, _vp, _rm, _vp, _rm, _i, _i, _i, _i, _vp, _sz, _vp]),
    'xps_gemm_tn_grouped_f32_workspace': (_sz, [_vp, _i]),
    'xps_gemm_tn_grouped_f32': (_i, [_vp, _i, _vp, _sz, _vp]),
    'xps_colsum_f32_workspace': (_sz, [_i, _i]),
    'xps_colsum_f32': (_i, [_vp, _i64, _i, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    'xps_gru_seq_fwd_f32_workspace': (_sz, [_i, _i, _i, _i]),
    'xps_gru_seq_status_offset': (C.c_longlong, [_i, _i, _i, _i]),
    'xps_gru_set_status_word': (_i, [_vp]),
    'xps_set_gru_bptt_grid': (_i, [_i]),
    'xps_get_gru_bptt_grid': (_i, []),
    'xps_set_gru_cluster_mode': (_i, [_i]),
    'xps_get_gru_cluster_mode': (_i, []),
    'xps_gru_seq_fwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'xps_gru_seq_bwd_f32_workspace': (_sz, [_i, _i, _i, _i]),
    'xps_gru_seq_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'xps_gru_seq_fused_dropout_supported': (_i, [_i, _i, _i, _i]),
    'xps_gru_seq_fwd_images_supported': (_i, [_i, _i, _i, _i]),
    'xps_gru_seq_fwd_image_exchange_supported': (_i, [_i, _i, _i, _i]),
    'xps_gru_seq_fwd_images_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, C.c_uint64, _vp, _sz, _vp]),
    'xps_gru_seq_fwd_drop_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _f, C.c_uint64, _vp, _sz, _vp]),
    'xps_gru_seq_bwd_drop_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, C.c_uint64, _vp, _sz, _vp]),
    'xps_gru_seq_bwd_split4_supported': (_i, [_i, _i, _i, _i]),
    'xps_gru_seq_bwd_split4_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, C.c_uint64, _vp, _sz, _vp]),
    'xps_transpose_f32': (_i, [_vp, _vp, _i, _i, _vp]),
    'xps_transpose_batched_f32': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'xps_bn_finalize_f32': (_i, [_vp, _d, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp]),
    'xps_bn_apply_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i64, _i, _i, _vp]),
    'xps_bn_finalize_apply_f32': (_i, [_vp, _vp, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _f, _vp, _i64, _i, _i, _vp]),
    'xps_bn_apply_eval_f32': (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    'xps_bn_bwd_workspace': (_sz, [_i64, _i]),
    'xps_bn_bwd_reduce_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _i64, _i, _vp, _sz, _vp]),
    'xps_bn_bwd_apply_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _d, _vp, _i64, _i, _vp]),
    'xps_decoder_supported': (_i, [_i, _i, _i]),
    'xps_decoder_fwd_f32': (_i, [_vp] * 12 + [_i] * 6 + [_vp]),
    'xps_decoder_bwd_f32': (_i, [_vp] * 8 + [_i] * 4 + [_vp]),
    'xps_gemv_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'xps_gru_cell_gemv_f32': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'xps_gather_rows_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'xps_scatter_rows_f32_workspace': (_sz, [_i, _i, _i]),
    'xps_scatter_rows_f32': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'xps_next_token': (_i, [_vp, _i, _vp, _i64, _vp, _vp, _i, _vp]),
    'xps_decoder_select_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'xps_dropout_f32': (_i, [_vp, _vp, _vp, _i64, _f, C.c_uint64, _vp]),
    'xps_split4_f32': (_i, [_vp, _vp, _i64, _f, C.c_uint64, _vp]),
    'xps_split4_pad_f32': (_i, [_vp, _i64, _i, _i, _vp, _i64, _vp]),
    'xps_mask_scale_f32': (_i, [_vp, _vp, _f, _vp, _i64, _vp]),
    'xps_add_f32': (_i, [_vp, _vp, _vp, _i64, _vp]),
    'xps_cross_entropy_fwd_f32': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp]),
    'xps_cross_entropy_bwd_f32': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp]),
    'xps_cross_entropy_loss_grad_f32_workspace': (_sz, [_i64]),
    'xps_cross_entropy_loss_grad_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i64, _i, _vp]),
    'xps_ctc_loss_f32_workspace': (_sz, [_i, _i, _i]),
    'xps_ctc_loss_f32': (_i, [_vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'xps_sumsq_f32_workspace': (_sz, [_i64]),
    'xps_sumsq_f32': (_i, [_vp, _i64, _vp, _vp, _sz, _vp]),
    'xps_adamw_f32': (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _i, _vp]),
    'xps_clip_adamw_f32': (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _i, _vp, _sz, _vp]),
    'xps_cnd_avg_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp]),
    'xps_cnd_avg_f64': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp]),
    'xps_colsum_f64_workspace': (_sz, [_i64, _i]),
    'xps_colsum_f64': (_i, [_vp, _i, _i64, _i64, _i, _vp, _vp, _sz, _vp]),
    'xps_xcov_f64_workspace': (_sz, [_i64, _i, _i]),
    'xps_xcov_f64': (_i, [_vp, _i, _i64, _vp, _vp, _i, _i64, _vp, _vp, _i64, _i64, _i, _i, _vp, _sz, _vp]),
    'xps_jacobi_f64_workspace': (_sz, [_i]),
    'xps_jacobi_sweeps_f64': (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'xps_jacobi_small_supported': (_i, [_i, _i, _i]),
    'xps_chol_whiten_supported': (_i, [_i]),
    'xps_chol_whiten_f64': (_i, [_vp, _i64, _i64, _d, _d, _vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _vp, _vp]),
    'xps_jacobi_small_f64': (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _d, _vp, _vp, _vp]),
    'xps_apply_f64': (_i, [_vp, _i, _i64, _vp, _vp, _i64, _vp, _i, _i64, _i64, _i, _i, _vp]),
    'xps_process_hg_f64_workspace': (_sz, [_i, _i, _i]),
    'xps_process_hg_f64': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'xps_aug_time_shift_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'xps_aug_time_mask_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'xps_aug_scale_f32': (_i, [_vp, _vp, _i64, _f, _vp]),
    'xps_aug_jitter_f32': (_i, [_vp, _vp, _vp, _i64, _f, _vp]),
    'xps_aug_time_warp_f32': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'xps_rbf_from_gram_f64': (_i, [_vp, _i64, _vp, _vp, _i, _i, _d, _vp, _i64, _vp]),
    'xps_svm_smo_f64_max_points': (_sz, []),
    'xps_svm_smo_f64': (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _vp, _d, _i, _vp, _vp, _vp, _vp]),
    'xps_dgemm_small': (_i, [_vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _i, _i, _vp]),
    'xps_dgemm_splitk_workspace': (_sz, [_i, _i, _i]),
    'xps_dgemm_splitk': (_i, [_vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _i, _i, _i, _vp, _sz, _vp]),
    'xps_lanczos_f64_workspace': (_sz, [_i]),
    'xps_lanczos_f64': (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'xps_cheb_filter_f64_workspace': (_sz, [_i, _i]),
    'xps_cheb_filter_f64': (_i, [_vp, _i64, _i, _vp, _i, _i, _d, _d, _d, _vp, _vp, _sz, _vp]),
}


def header_functions(path=HEADER_PATH):
    """Names of all functions declared in include/xps.h."""
    with open(path) as f:
        src = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(xps_[a-z0-9_]+)\s*\(', src)))


def lib():
    """Load libxps.so once.  Raises if it is not built — there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise XpsError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(hipcc --offload-arch=gfx950).  The HIP path has no CPU fallback.')
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as the system
        # one).  It must be loaded FIRST so that libxps.so binds to the runtime that owns torch's
        # device context and streams; two runtimes in one process see "no ROCm-capable device".
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().xps_last_error()
        raise XpsError(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')


_fns = {}


def call(name, *args):
    """Call an int-returning entry point and raise on a non-zero code.  (The bound functions are cached: a training step makes
    ~45 of these calls and the host path is as long as the GPU's.)"""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc:
        check(rc, name)
