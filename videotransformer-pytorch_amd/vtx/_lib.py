"""ctypes binding of libvtx.so (include/vtx.h).  No torch types cross this boundary:
only raw device pointers, sizes and the HIP stream handle.

The library is looked up next to this package (videotransformer-pytorch_amd/
libvtx.so, built by csrc/build.py).  There is NO fallback: if it is missing or a
symbol is absent, importing the ops raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
LIB_PATH = os.environ.get('VTX_LIB', os.path.join(_PKG, 'libvtx.so'))

VTX_F32, VTX_BF16, VTX_BF16_X32 = 0, 1, 3
ATTN_CONTIG, ATTN_SPACE = 0, 1


class RowMap(C.Structure):
    _fields_ = [('grp', C.c_int), ('skip', C.c_int), ('base', C.c_int), ('tab', C.c_void_p)]


IDENT = RowMap(0, 0, 0, None)


class GemmDesc(C.Structure):
    _fields_ = [
        ('dtype', C.c_int), ('M', C.c_int), ('N', C.c_int), ('K', C.c_int),
        ('A', C.c_void_p), ('lda', C.c_long), ('amap', RowMap),
        ('B', C.c_void_p), ('ldb', C.c_long),
        ('C', C.c_void_p), ('ldc', C.c_long), ('cmap', RowMap),
        ('bias', C.c_void_p),
        ('act', C.c_int), ('C2', C.c_void_p), ('ldc2', C.c_long),
        ('dgelu_in', C.c_void_p), ('ld_dgelu', C.c_long), ('dgelu_kind', C.c_int),
        ('row_scale', C.c_void_p),
        ('rs_d1', C.c_int), ('rs_m1', C.c_int), ('rs_d2', C.c_int), ('rs_m2', C.c_int),
        ('R', C.c_void_p), ('ldr', C.c_long), ('rmap', RowMap), ('r_period', C.c_int),
        ('split_row', C.c_int), ('Csplit', C.c_void_p), ('ldsplit', C.c_long),
        ('workspace', C.c_void_p), ('ws_bytes', C.c_size_t),
    ]


class GemmTnDesc(C.Structure):
    _fields_ = [
        ('dtype', C.c_int), ('M', C.c_int), ('N1', C.c_int), ('N2', C.c_int),
        ('A', C.c_void_p), ('lda', C.c_long), ('amap', RowMap),
        ('B', C.c_void_p), ('ldb', C.c_long), ('bmap', RowMap),
        ('C', C.c_void_p), ('ldc', C.c_long), ('accumulate', C.c_int),
        ('workspace', C.c_void_p), ('ws_bytes', C.c_size_t),
        ('colsum', C.c_void_p), ('colsum_accumulate', C.c_int),
    ]


class WprodDesc(C.Structure):
    _fields_ = [
        ('N1', C.c_int), ('N2', C.c_int), ('K', C.c_int),
        ('A', C.c_void_p), ('a_rs', C.c_long), ('a_ks', C.c_long),
        ('B', C.c_void_p), ('b_ks', C.c_long), ('b_cs', C.c_long),
        ('alpha', C.c_float),
        ('C', C.c_void_p), ('ldc', C.c_long), ('accumulate', C.c_int),
        ('u', C.c_void_p), ('v', C.c_void_p),
        ('x', C.c_void_p), ('y', C.c_void_p),
        ('alpha_y', C.c_float), ('z', C.c_void_p), ('beta_z', C.c_float), ('y_accumulate', C.c_int),
    ]


class CtTensor(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst_c', C.c_void_p), ('dst_t', C.c_void_p), ('rows', C.c_int), ('cols', C.c_int)]


class MtTensor(C.Structure):
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('s1', C.c_void_p), ('s2', C.c_void_p), ('n', C.c_long),
                ('lr', C.c_float), ('wd', C.c_float)]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('dtype', 'B', 'T', 'H', 'W', 'heads', 'hd', 'sh', 'sw')]


class XAttnDesc(C.Structure):
    _fields_ = [('dtype', C.c_int), ('B', C.c_int), ('Lq', C.c_int), ('Lk', C.c_int), ('heads', C.c_int), ('hd', C.c_int),
                ('scale', C.c_float), ('q', C.c_void_p), ('k', C.c_void_p), ('v', C.c_void_p), ('out', C.c_void_p),
                ('lse', C.c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [
        ('dtype', C.c_int), ('mode', C.c_int),
        ('S', C.c_int), ('L', C.c_int), ('H', C.c_int), ('hd', C.c_int),
        ('B', C.c_int), ('T', C.c_int), ('P', C.c_int),
        ('qkv', C.c_void_p), ('ld_qkv', C.c_long),
        ('out', C.c_void_p), ('ld_out', C.c_long),
        ('lse', C.c_void_p), ('probs', C.c_void_p), ('scale', C.c_float),
    ]


class AttnBwdDesc(C.Structure):
    _fields_ = [
        ('f', AttnDesc),
        ('dout', C.c_void_p), ('ld_dout', C.c_long),
        ('dqkv', C.c_void_p), ('ld_dqkv', C.c_long),
        ('dqkv_cls', C.c_void_p), ('delta', C.c_void_p),
    ]


vp, ci, cl, cf, sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# name -> (restype, argtypes).  Must list every symbol include/vtx.h declares.
SIGNATURES = {
    'vtx_version': (ci, []),
    'vtx_last_error_string': (C.c_char_p, []),
    'vtx_set_option': (ci, [C.c_char_p, C.c_char_p]),
    'vtx_layernorm_fwd': (ci, [ci, ci, ci, vp, cl, RowMap, vp, vp, cf, vp, cl, RowMap, vp, vp, vp]),
    'vtx_layernorm_acc_fwd': (ci, [ci, ci, vp, vp, cl, RowMap, vp, cl, RowMap, vp, vp, cf, vp, cl, RowMap, vp, vp, vp]),
    'vtx_layernorm_bwd_workspace': (sz, [ci, ci]),
    'vtx_layernorm_bwd': (ci, [ci, ci, ci, vp, cl, RowMap, vp, cl, RowMap, vp, vp, vp, vp, vp, cl,
                               vp, vp, vp, sz, vp]),
    'vtx_layernorm_bwd_g32': (ci, [ci, ci, vp, cl, RowMap, vp, cl, RowMap, vp, vp, vp, vp, vp, vp, cl, vp, vp, vp, sz, vp]),
    'vtx_gemm_nt_workspace': (sz, []),
    'vtx_gemm_nt': (ci, [C.POINTER(GemmDesc), vp]),
    'vtx_gemm_tn_workspace': (sz, [ci, ci, ci]),
    'vtx_gemm_tn': (ci, [C.POINTER(GemmTnDesc), vp]),
    'vtx_wprod': (ci, [C.POINTER(WprodDesc), vp]),
    'vtx_colsum_workspace': (sz, [ci, ci]),
    'vtx_colsum': (ci, [ci, ci, ci, vp, cl, RowMap, vp, ci, vp, sz, vp]),
    'vtx_attn_fwd': (ci, [C.POINTER(AttnDesc), vp]),
    'vtx_attn_bwd': (ci, [C.POINTER(AttnBwdDesc), vp]),
    'vtx_cls_mean_fwd': (ci, [ci, ci, ci, ci, vp, cl, vp, vp, cl, cl, vp]),
    'vtx_fact_glue_fwd': (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp]),
    'vtx_fact_glue_bwd': (ci, [ci, ci, ci, ci, ci, vp, vp, vp, ci, vp]),
    'vtx_space_grad_prep': (ci, [ci, ci, ci, ci, ci, vp, cl, vp, vp, cl, vp]),
    'vtx_cls_qkv_reduce': (ci, [ci, ci, ci, ci, vp, cl, vp, cl, cl, vp]),
    'vtx_dropped_rows_fix': (ci, [ci, cl, ci, ci, vp, vp, cl, RowMap, vp, vp, cl, RowMap, vp, cl, vp]),
    'vtx_dropped_rows_colsum': (ci, [ci, cl, ci, ci, vp, vp, cl, RowMap, vp, ci, vp]),
    'vtx_row_scale_copy': (ci, [ci, ci, ci, vp, cl, RowMap, vp, cl, RowMap, vp, ci, ci, ci, ci, vp]),
    'vtx_reduce_rows': (ci, [ci, ci, ci, ci, vp, cl, cl, cl, cl, vp, cl, cf, ci, vp]),
    'vtx_gelu_grad_mul': (ci, [ci, sz, vp, vp, vp, vp]),
    'vtx_cast_transpose': (ci, [ci, ci, ci, vp, vp, vp, vp]),
    'vtx_mt_cast_transpose': (ci, [ci, vp, vp, ci, ci, vp]),
    'vtx_cast_from_f32': (ci, [ci, sz, vp, vp, vp]),
    'vtx_cast_to_f32': (ci, [ci, sz, vp, vp, vp]),
    'vtx_patch_rows': (ci, [ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, cl, ci, vp]),
    'vtx_patch_rows_u8': (ci, [ci, ci, ci, ci, ci, ci, ci, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, cl, ci, vp]),
    'vtx_embed_table': (ci, [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, vp]),
    'vtx_hog_table_bytes': (sz, []),
    'vtx_hog_build_table': (ci, [vp]),
    'vtx_hog_fwd': (ci, [vp, ci, ci, ci, vp, sz, vp, vp, vp]),
    'vtx_maskfeat_blend_fwd': (ci, [ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp]),
    'vtx_maskfeat_blend_bwd': (ci, [ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp]),
    'vtx_maskfeat_loss_fwd': (ci, [ci, ci, ci, ci, ci, ci, vp, cl, vp, vp, vp, vp]),
    'vtx_maskfeat_loss_bwd': (ci, [ci, ci, ci, ci, ci, ci, vp, cl, vp, vp, vp, cf, vp, cl, vp]),
    'vtx_pool_conv_ln_fwd': (ci, [C.POINTER(PoolDesc), vp, vp, vp, vp, cf, vp, vp, vp, vp, vp]),
    'vtx_pool_conv_ln_bwd_workspace': (sz, [C.POINTER(PoolDesc)]),
    'vtx_pool_conv_ln_bwd': (ci, [C.POINTER(PoolDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    'vtx_maxpool_skip_fwd': (ci, [ci, ci, ci, ci, ci, ci, vp, vp, vp, vp]),
    'vtx_maxpool_skip_bwd': (ci, [ci, ci, ci, ci, ci, ci, vp, vp, vp, vp]),
    'vtx_pos_encoding_fwd': (ci, [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]),
    'vtx_im2col3d': (ci, [ci, ci, ci, ci, ci, ci, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), ci, vp, vp, vp]),
    'vtx_xattn_fwd': (ci, [C.POINTER(XAttnDesc), vp]),
    'vtx_xattn_bwd_workspace': (sz, [C.POINTER(XAttnDesc)]),
    'vtx_xattn_bwd': (ci, [C.POINTER(XAttnDesc), vp, vp, vp, vp, vp, vp, sz, vp]),
    'vtx_mixup_batch': (ci, [vp, ci, cl, cf, cf, vp]),
    'vtx_cutmix_batch': (ci, [vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]),
    'vtx_mixup_target': (ci, [vp, ci, ci, cf, cf, cf, cf, vp, vp]),
    'vtx_softmax_xent_fwd': (ci, [vp, vp, vp, ci, ci, vp, vp, vp, vp]),
    'vtx_softmax_xent_bwd': (ci, [vp, vp, vp, vp, ci, ci, cf, vp, vp, vp, vp]),
    'vtx_topk_correct': (ci, [vp, vp, ci, ci, ci, vp, vp]),
    'vtx_mt_chunks': (ci, [cl]),
    'vtx_mt_grad_norms': (ci, [vp, vp, ci, ci, vp, vp, vp]),
    'vtx_mt_sgd_step': (ci, [vp, vp, ci, ci, vp, cf, cf, ci, ci, vp]),
    'vtx_mt_adamw_step': (ci, [vp, vp, ci, ci, vp, cf, cf, cf, cf, ci, vp]),
    'vtx_selftest': (ci, [C.c_char_p, sz]),
}

_lib = None


class VtxError(RuntimeError):
    pass


def load():
    """Load libvtx.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise VtxError(
            f'libvtx.so not found at {LIB_PATH}: build it with '
            f'`python videotransformer-pytorch_amd/csrc/build.py` (there is no CPU/PyTorch fallback)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VtxError(f'libvtx.so does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().vtx_last_error_string()
        raise VtxError(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')


def set_option(name, value):
    """Tuning / diagnostic switch of the library (see vtx_set_option in include/vtx.h)."""
    rc = load().vtx_set_option(str(name).encode(), str(value).encode())
    if rc != 0:
        check(rc, f'vtx_set_option({name}={value})')


_TRACE = os.environ.get('VTX_TRACE_CALLS', '0') == '1'


def call(name, *args):
    if _TRACE:                      # debugging aid: name every launch and wait for it (finds the faulting kernel)
        import sys
        import torch
        sys.stderr.write(f'[vtx] {name}\n')
        sys.stderr.flush()
    rc = getattr(load(), name)(*args)
    if rc != 0:
        check(rc, name)
    if _TRACE:
        torch.cuda.synchronize()
