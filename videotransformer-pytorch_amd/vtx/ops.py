"""Tensor-level wrappers over the C ABI (one Python function per libvtx entry point).

PyTorch is used here only for device memory (``torch.empty``), the current HIP
stream and dtype bookkeeping; every FLOP on the path is issued by libvtx.so.
All tensors must be CUDA (HIP) tensors; CPU tensors raise -- there is no fallback.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import RowMap, IDENT, GemmDesc, GemmTnDesc, AttnDesc, AttnBwdDesc, call  # noqa: F401

_DT = {torch.float32: _lib.VTX_F32, torch.bfloat16: _lib.VTX_BF16}


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f'vtx: unsupported activation dtype {t.dtype} (float32 or bfloat16)')


def need_cuda(*ts):
    """Every operand must live on the CURRENT HIP device (kernels are launched on its current stream)."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('vtx: the HIP path needs CUDA/HIP tensors (no CPU fallback); '
                               f'got a tensor on {t.device}')
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(f'vtx: tensor on {t.device} but the current device is cuda:{cur} '
                               '(wrap the call in torch.cuda.device(...))')


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def rowmap(grp=0, skip=0, base=0):
    return RowMap(int(grp), int(skip), int(base))


def tabmap(rows_per_group, table, max_step):
    """Table row map: logical row m -> m + table[m // rows_per_group] (``table``: int32 device tensor, one row offset per
    group plus a spare entry; ``max_step``: upper bound of the difference of two consecutive entries)."""
    need_cuda(table)
    if table.dtype != torch.int32:
        raise TypeError('tabmap: int32 table expected')
    return RowMap(int(rows_per_group), int(max_step), 0, table.data_ptr())


def upload_i32(values, device):
    """list / array of ints -> int32 device tensor, through the pinned-memory copy kernel of upload_f32 (bit copy)."""
    import numpy as np
    a = np.asarray(values, dtype=np.int32)
    return upload_f32(torch.from_numpy(a.view(np.float32)), device).view(torch.int32)


def tokmap(n_tokens):
    """logical token row (clip-major, no cls) -> physical row of a [B, 1+N, D] tensor."""
    return RowMap(int(n_tokens), 1, 1)


def clsmap(n_tokens):
    """logical row b -> physical row b*(1+N): the cls row of every clip."""
    return RowMap(1, int(n_tokens), 0)


# ---- optional per-launch timing (bench.py rooflines): HIP events on the launch stream ----
_prof = None


def profile_start(classes=('gemm_nt',)):
    """Time every launch of the named kernel classes with a pair of HIP events on the launch stream
    (torch.cuda.Event records on torch's current stream, which is the stream the kernels are launched on)."""
    global _prof
    _prof = {c: [] for c in classes}


def profile_stop():
    """-> {class: {shape_key: [launches, total_ms, flops, algorithmic_bytes]}}; call after torch.cuda.synchronize()."""
    global _prof
    out = {}
    for c, recs in (_prof or {}).items():
        per = out.setdefault(c, {})
        for e0, e1, key, fl, by in recs:
            r = per.setdefault(key, [0, 0.0, 0.0, 0.0])
            r[0] += 1
            r[1] += e0.elapsed_time(e1)
            r[2] += fl
            r[3] += by
    _prof = None
    return out


def profile_totals(per_class):
    """{class: {key: [n, ms, flops, bytes]}} -> {class: (n, ms, flops, bytes)}."""
    return {c: tuple(sum(v[i] for v in d.values()) for i in range(4)) for c, d in per_class.items()}


class _timed:
    def __init__(self, cls, flops=0.0, nbytes=0.0, key=''):
        self.on = _prof is not None and cls in _prof
        self.cls, self.rec = cls, (key, float(flops), float(nbytes))

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            _prof[self.cls].append((self.e0, self.e1) + self.rec)


def _f32(t):
    if t is not None and t.dtype != torch.float32:
        raise TypeError('vtx: parameter tensors must be float32')
    return t


# ------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, rows, D, ldx, xmap, gamma, beta, eps, y, ldy, ymap=IDENT, mean=None, rstd=None):
    need_cuda(x, y, gamma, beta)
    with _timed('ln_fwd', nbytes=rows * D * (x.element_size() + y.element_size()), key=f'{rows}x{D}'):
        call('vtx_layernorm_fwd', dt(x), rows, D, ptr(x), ldx, xmap, ptr(_f32(gamma)), ptr(_f32(beta)), float(eps),
             ptr(y), ldy, ymap, ptr(mean), ptr(rstd), stream())


def layernorm_acc_fwd(xs, d, rows, D, lds, smap, xo, ldo, omap, gamma=None, beta=None, eps=1e-5, y=None, ldy=0, ymap=IDENT,
                      mean=None, rstd=None):
    """The exact residual stream (vtx_layernorm_acc_fwd): xo = (xs or 0) + d in float32 on the mapped rows and, with ``y``, the
    LayerNorm of those rows (bf16).  xs: float32 stream or None; d: bf16 contribution in the stream layout."""
    need_cuda(d, xo, xs, y)
    if d.dtype != torch.bfloat16 or xo.dtype != torch.float32 or (xs is not None and xs.dtype != torch.float32):
        raise TypeError('layernorm_acc_fwd: d bfloat16, xs / xo float32')
    nb = rows * D * (2 + 4 + (4 if xs is not None else 0) + (2 if y is not None else 0))
    with _timed('ln_fwd', nbytes=nb, key=f'acc {rows}x{D}'):
        call('vtx_layernorm_acc_fwd', rows, D, ptr(xs), ptr(d), lds, smap, ptr(xo), ldo, omap, ptr(_f32(gamma)), ptr(_f32(beta)),
             float(eps), ptr(y), ldy, ymap, ptr(mean), ptr(rstd), stream())


def layernorm_bwd(dy, lddy, dymap, x, ldx, xmap, rows, D, mean, rstd, gamma, dres, dx, lddx, dgamma, dbeta, dres32=None, dx32=None):
    """dx = (dres or 0) + LN'(dy); with dres32 / dx32 (float32 gradient stream, bf16 kernels over the float32 stream x):
    dx32 = dres32 + LN'(dy) in float32, dx = bf16(dx32) (vtx_layernorm_bwd_g32; ``dres`` is not read)."""
    need_cuda(dy, x, dx)
    ws_bytes = _lib.load().vtx_layernorm_bwd_workspace(rows, D)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    if dres32 is not None:
        need_cuda(dres32, dx32)
        if dy.dtype != torch.bfloat16 or x.dtype != torch.float32 or dres32.dtype != torch.float32 or dx32.dtype != torch.float32:
            raise TypeError('layernorm_bwd: the float32 gradient stream takes bf16 dy / dx and float32 x / dres32 / dx32')
        with _timed('ln_bwd', nbytes=(2 + 4 + 4 + 4 + 2) * rows * D, key=f'g32 {rows}x{D}'):
            call('vtx_layernorm_bwd_g32', rows, D, ptr(dy), lddy, dymap, ptr(x), ldx, xmap, ptr(mean), ptr(rstd), ptr(gamma), ptr(dres32),
                 ptr(dx32), ptr(dx), lddx, ptr(dgamma), ptr(dbeta), ptr(ws), ws_bytes, stream())
        return
    # bf16 gradients with x stored as float32: the exact residual stream (vtx.set_stream('fp32'))
    mixed = dy.dtype == torch.bfloat16 and x.dtype == torch.float32
    with _timed('ln_bwd', nbytes=((3.0 if dres is not None else 2.0) * dy.element_size() + x.element_size()) * rows * D, key=f'{rows}x{D}'):
        call('vtx_layernorm_bwd', _lib.VTX_BF16_X32 if mixed else dt(dy), rows, D, ptr(dy), lddy, dymap, ptr(x), ldx, xmap, ptr(mean), ptr(rstd),
             ptr(gamma), ptr(dres), ptr(dx), lddx, ptr(dgamma), ptr(dbeta), ptr(ws), ws_bytes, stream())


# ----------------------------------------------------------------------- GEMM
_nt_ws = {}


def _gemm_nt_workspace(device):
    """Tile counters of the persistent GEMM: one zeroed buffer per (device, stream); the kernel leaves it
    zeroed, and launches on one stream are ordered, so it is never shared by two launches in flight."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _nt_ws.get(key)
    if ws is None:
        ws = torch.zeros(_lib.load().vtx_gemm_nt_workspace() // 4, dtype=torch.int32, device=device)
        _nt_ws[key] = ws
    return ws


def gemm_nt(A, B, Cout, M, N, K, lda=None, ldb=None, ldc=None, amap=IDENT, cmap=IDENT, bias=None, act=0,
            C2=None, dgelu_in=None, dgelu_kind=0, row_scale=None, rs=(1, 0, 1, 0), R=None, ldr=None, rmap=IDENT,
            r_period=0, split_row=0, Csplit=None):
    """C = epilogue(A[M,K] @ B[N,K]^T); see vtx_gemm_nt in include/vtx.h."""
    need_cuda(A, B, Cout)
    if A.dtype != B.dtype or A.dtype != Cout.dtype:
        raise TypeError(f'gemm_nt: dtype mismatch {A.dtype} {B.dtype} {Cout.dtype}')
    d = GemmDesc()
    d.dtype = dt(A); d.M, d.N, d.K = int(M), int(N), int(K)
    d.A = ptr(A); d.lda = K if lda is None else lda; d.amap = amap
    d.B = ptr(B); d.ldb = K if ldb is None else ldb
    d.C = ptr(Cout); d.ldc = N if ldc is None else ldc; d.cmap = cmap
    d.bias = ptr(_f32(bias)); d.act = act
    d.C2 = ptr(C2); d.ldc2 = N
    d.dgelu_in = ptr(dgelu_in); d.ld_dgelu = N; d.dgelu_kind = int(dgelu_kind)
    d.row_scale = ptr(_f32(row_scale)); d.rs_d1, d.rs_m1, d.rs_d2, d.rs_m2 = [int(v) for v in rs]
    d.R = ptr(R); d.ldr = (N if ldr is None else ldr); d.rmap = rmap; d.r_period = int(r_period)
    d.split_row = int(split_row); d.Csplit = ptr(Csplit); d.ldsplit = N
    ws = _gemm_nt_workspace(A.device)
    d.workspace = ptr(ws); d.ws_bytes = ws.numel() * 4
    if _lib._TRACE:
        import sys
        sys.stderr.write(f'[vtx]   gemm_nt M={M} N={N} K={K} lda={d.lda} ldc={d.ldc} A={tuple(A.shape)} B={tuple(B.shape)} C={tuple(Cout.shape)} amap=({amap.grp},{amap.skip},{amap.base}) cmap=({cmap.grp},{cmap.skip},{cmap.base}) bias={bias is not None} rs={row_scale is not None} R={R is not None} split={split_row}\n')
    if _prof is not None and 'gemm_nt' in _prof:
        # algorithmic bytes: every operand the launch must touch once (A, weight, C, and what the epilogue
        # reads or writes besides: residual, GELU' input, pre-activation copy)
        es = A.element_size()
        nb = (M * K + N * K + M * N) * es
        tags = ''
        if C2 is not None:
            nb += M * N * es
            tags += '+preact' if act == 1 else "+gelu'out"
        if dgelu_in is not None:
            nb += M * N * es
            tags += "+gelu'" if dgelu_kind == 0 else '+mul'
        if R is not None:
            nb += (min(M, r_period) if r_period else M) * N * es
            tags += '+res'
        if row_scale is not None:
            tags += '+scale'
        with _timed('gemm_nt', 2.0 * M * N * K, nb, f'{M}x{N}x{K}{tags}'):
            call('vtx_gemm_nt', C.byref(d), stream())
    else:
        call('vtx_gemm_nt', C.byref(d), stream())


def gemm_tn(A, B, M, N1, N2, out=None, lda=None, ldb=None, amap=IDENT, bmap=IDENT, accumulate=False,
            want_colsum=False, colsum_out=None, colsum_accumulate=False):
    """out[N1,N2] (fp32) (+)= sum_m A[m,:N1]^T B[m,:N2]; with want_colsum (or a colsum_out buffer, optionally
    accumulated into) also returns sum_m A[m,:N1] (the bias gradient that always accompanies a weight gradient)."""
    need_cuda(A, B)
    if out is None:
        out = torch.empty(N1, N2, dtype=torch.float32, device=A.device)
    if colsum_out is not None:
        need_cuda(colsum_out)
        want_colsum = True
        cs = colsum_out
    else:
        cs = torch.empty(N1, dtype=torch.float32, device=A.device) if want_colsum else None
    ws_bytes = _lib.load().vtx_gemm_tn_workspace(M, N1, N2)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=A.device)
    d = GemmTnDesc()
    d.dtype = dt(A); d.M, d.N1, d.N2 = int(M), int(N1), int(N2)
    d.A = ptr(A); d.lda = N1 if lda is None else lda; d.amap = amap
    d.B = ptr(B); d.ldb = N2 if ldb is None else ldb; d.bmap = bmap
    d.C = ptr(out); d.ldc = N2; d.accumulate = int(bool(accumulate))
    d.workspace = ptr(ws); d.ws_bytes = ws_bytes
    d.colsum = ptr(cs); d.colsum_accumulate = int(bool(colsum_accumulate and colsum_out is not None))
    with _timed('gemm_tn', 2.0 * M * N1 * N2, M * (N1 + N2) * A.element_size() + N1 * N2 * 4, f'{M}x{N1}x{N2}'):
        call('vtx_gemm_tn', C.byref(d), stream())
    return (out, cs) if want_colsum else out


def wprod(A, B, ta=False, tb=False, out=None, alpha=1.0, accumulate=False, u=None, v=None, x=None, y=None, alpha_y=1.0,
          z=None, beta_z=0.0, y_accumulate=False):
    """out (+)= alpha * op(A) @ op(B) (+ u v^T) on contiguous fp32 matrices, op = transpose when ta / tb; with x also
    y (+)= alpha_y * op(A) @ x + beta_z * z (y is allocated when None).  Returns out, or (out, y).  See vtx_wprod."""
    need_cuda(A, B, out, u, v, x, y, z)
    for t in (A, B, out, u, v, x, y, z):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise TypeError('wprod: contiguous float32 tensors expected')
    d = _lib.WprodDesc()
    ar, ac = A.shape
    br, bc = B.shape
    if ta:
        d.N1, K, d.a_rs, d.a_ks = ac, ar, 1, ac
    else:
        d.N1, K, d.a_rs, d.a_ks = ar, ac, ac, 1
    if tb:
        Kb, d.N2, d.b_ks, d.b_cs = bc, br, 1, bc
    else:
        Kb, d.N2, d.b_ks, d.b_cs = br, bc, bc, 1
    if K != Kb:
        raise ValueError(f'wprod: inner dimensions differ ({K} vs {Kb})')
    d.K = K
    if out is None:
        if accumulate:
            raise ValueError('wprod: accumulate needs an output tensor')
        out = torch.empty(d.N1, d.N2, dtype=torch.float32, device=A.device)
    if x is not None and y is None:
        if y_accumulate:
            raise ValueError('wprod: y_accumulate needs y')
        y = torch.empty(d.N1, dtype=torch.float32, device=A.device)
    d.A, d.B, d.C, d.ldc = ptr(A), ptr(B), ptr(out), d.N2
    d.alpha, d.accumulate = float(alpha), int(bool(accumulate))
    d.u, d.v, d.x, d.y, d.z = ptr(u), ptr(v), ptr(x), ptr(y), ptr(z)
    d.alpha_y, d.beta_z, d.y_accumulate = float(alpha_y), float(beta_z), int(bool(y_accumulate))
    with _timed('wprod', 2.0 * d.N1 * d.N2 * K, 4.0 * (d.N1 * K + K * d.N2 + d.N1 * d.N2), f'{d.N1}x{d.N2}x{K}'):
        call('vtx_wprod', C.byref(d), stream())
    return out if x is None else (out, y)


def colsum(A, M, N, lda=None, amap=IDENT, out=None, accumulate=False):
    need_cuda(A)
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=A.device)
    ws_bytes = _lib.load().vtx_colsum_workspace(M, N)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=A.device)
    with _timed('colsum', nbytes=1.0 * M * N * A.element_size(), key=f'{M}x{N}'):
        call('vtx_colsum', dt(A), M, N, ptr(A), N if lda is None else lda, amap, ptr(out), int(bool(accumulate)),
             ptr(ws), ws_bytes, stream())
    return out


# ------------------------------------------------------------------ attention
def _attn_desc(qkv, out, lse, mode, S, L, H, hd, scale, B=0, T=0, P=0, probs=None):
    d = AttnDesc()
    d.dtype = dt(qkv); d.mode = mode
    d.S, d.L, d.H, d.hd = int(S), int(L), int(H), int(hd)
    d.B, d.T, d.P = int(B), int(T), int(P)
    d.qkv = ptr(qkv); d.ld_qkv = 3 * H * hd
    d.out = ptr(out); d.ld_out = H * hd
    d.lse = ptr(lse); d.probs = ptr(probs); d.scale = float(scale)
    return d


def attn_fwd(qkv, out, lse, mode, S, L, H, hd, scale, B=0, T=0, P=0, probs=None):
    need_cuda(qkv, out, lse)
    d = _attn_desc(qkv, out, lse, mode, S, L, H, hd, scale, B, T, P, probs)
    kind = 'space' if mode == _lib.ATTN_SPACE else ('time' if L <= 32 else 'seq')
    rows = S * L                       # algorithmic bytes: read q,k,v, write o (+ fp32 log-sum-exp)
    with _timed(f'attn_fwd_{kind}', 4.0 * S * H * L * L * hd, rows * H * hd * 4 * qkv.element_size() + rows * H * 4,
                f'{S}x{L}x{H}'):
        call('vtx_attn_fwd', C.byref(d), stream())


def attn_bwd(qkv, out, lse, dout, dqkv, mode, S, L, H, hd, scale, B=0, T=0, P=0, dqkv_cls=None):
    need_cuda(qkv, out, dout, dqkv)
    b = AttnBwdDesc()
    b.f = _attn_desc(qkv, out, lse, mode, S, L, H, hd, scale, B, T, P)
    b.dout = ptr(dout); b.ld_dout = H * hd
    b.dqkv = ptr(dqkv); b.ld_dqkv = 3 * H * hd
    b.dqkv_cls = ptr(dqkv_cls)
    delta = torch.empty(S * H * L, dtype=torch.float32, device=qkv.device)
    b.delta = ptr(delta)
    kind = 'space' if mode == _lib.ATTN_SPACE else ('time' if L <= 32 else 'seq')
    rows = S * L                       # read q,k,v,o,do, write dq,dk,dv
    with _timed(f'attn_bwd_{kind}', 10.0 * S * H * L * L * hd, rows * H * hd * 8 * qkv.element_size() + rows * H * 4,
                f'{S}x{L}x{H}'):
        call('vtx_attn_bwd', C.byref(b), stream())


# ------------------------------------------------------------------- glue ops
def fact_glue_fwd(x, time_embed, B, T, P, D):
    """ViViT fact_encoder glue: x [(B T), 1 + P, D] -> h [B, 1 + T, D] (vtx_fact_glue_fwd)."""
    need_cuda(x, time_embed)
    h = torch.empty(B, 1 + T, D, dtype=x.dtype, device=x.device)
    call('vtx_fact_glue_fwd', dt(x), B, T, P, D, ptr(x), ptr(_f32(time_embed)), ptr(h), stream())
    return h


def fact_glue_bwd(dh, B, T, P, D, d_time_embed=None, accumulate=False):
    need_cuda(dh)
    dx = torch.empty(B * T, 1 + P, D, dtype=dh.dtype, device=dh.device)
    call('vtx_fact_glue_bwd', dt(dh), B, T, P, D, ptr(dh), ptr(dx), ptr(d_time_embed), int(bool(accumulate)), stream())
    return dx


def cls_mean_fwd(a_cls, x, out, B, T, D, rows_per_clip):
    call('vtx_cls_mean_fwd', dt(out), B, T, D, ptr(a_cls), D, ptr(x), ptr(out), D, rows_per_clip, stream())     # x None: the mean alone


def space_grad_prep(dout, s, da, B, T, P, D):
    call('vtx_space_grad_prep', dt(dout), B, T, P, D, ptr(dout), D, ptr(s), ptr(da), D, stream())


def cls_qkv_reduce(dqkv_cls, dqkv, B, T, W, rows_per_clip):
    call('vtx_cls_qkv_reduce', dt(dqkv), B, T, W, ptr(dqkv_cls), W, ptr(dqkv), W, rows_per_clip, stream())


def row_scale_copy(src, dst, rows, D, lds=None, smap=IDENT, ldd=None, dmap=IDENT, s=None, rs=(1, 0, 1, 0)):
    need_cuda(src, dst)
    call('vtx_row_scale_copy', dt(src), rows, D, ptr(src), D if lds is None else lds, smap, ptr(dst),
         D if ldd is None else ldd, dmap, ptr(s), int(rs[0]), int(rs[1]), int(rs[2]), int(rs[3]), stream())


def dropped_rows_fix(s, M, D, group_rows, x=None, xmap=IDENT, bias=None, out=None, omap=IDENT, zero=None, ldx=None, ldo=None):
    """Rows of the groups with s == 0: out[omap(m)] = x[xmap(m)] + bias; zero[m] = 0 (either part optional)."""
    ref = out if out is not None else zero
    need_cuda(ref, s)
    call('vtx_dropped_rows_fix', dt(ref), M, D, group_rows, ptr(s), ptr(x), D if ldx is None else ldx, xmap, ptr(bias),
         ptr(out), D if ldo is None else ldo, omap, ptr(zero), D, stream())


def dropped_rows_colsum(src, s, M, D, group_rows, smap=IDENT, lds=None, nparts=96):
    """[nparts, D] fp32 partial column sums of src over the rows of the groups with s == 0."""
    need_cuda(src, s)
    part = torch.empty((nparts, D), dtype=torch.float32, device=src.device)
    call('vtx_dropped_rows_colsum', dt(src), M, D, group_rows, ptr(s), ptr(src), D if lds is None else lds, smap, ptr(part),
         nparts, stream())
    return part


def reduce_rows(inp, nj, ni, D, ld, base, si, sj, out=None, scale=1.0, accumulate=False):
    need_cuda(inp)
    if out is None:
        out = torch.empty(nj, D, dtype=torch.float32, device=inp.device)
    call('vtx_reduce_rows', dt(inp), nj, ni, D, ptr(inp), ld, base, si, sj, ptr(out), D, float(scale),
         int(bool(accumulate)), stream())
    return out


def cast_transpose(W, dtype, want_c=True, want_t=True):
    """fp32 [R,C] parameter -> (Wc [R,C], WcT [C,R]) in `dtype` (fp32: Wc is W itself)."""
    need_cuda(W)
    if W.dtype != torch.float32:
        raise TypeError(f'vtx: cast_transpose stages fp32 weights, got {W.dtype}')
    R, Cc = W.shape
    W = W.contiguous()
    code = _DT[dtype]
    Wc = None
    if want_c:
        Wc = W if dtype == torch.float32 else torch.empty(R, Cc, dtype=dtype, device=W.device)
    WcT = torch.empty(Cc, R, dtype=dtype, device=W.device) if want_t else None
    wc_arg = None if (Wc is None or Wc is W) else Wc
    if wc_arg is not None or WcT is not None:
        call('vtx_cast_transpose', code, R, Cc, ptr(W), ptr(wc_arg), ptr(WcT), stream())
    return Wc, WcT


def ct_table(entries, device):
    """Device table for mt_cast_transpose: entries = [(W fp32 [R,C] contiguous, Wc or None, WcT or None), ...].
    Returns (table tensor, tile prefix sums tensor, total tiles)."""
    tab = (_lib.CtTensor * len(entries))()
    starts = [0]
    for i, (W, wc, wt) in enumerate(entries):
        need_cuda(W)
        if not W.is_contiguous() or W.dtype != torch.float32:
            raise ValueError('ct_table: contiguous float32 matrices expected')
        R, Cc = W.shape
        tab[i].src, tab[i].dst_c, tab[i].dst_t = W.data_ptr(), ptr(wc), ptr(wt)
        tab[i].rows, tab[i].cols = R, Cc
        starts.append(starts[-1] + ((R + 63) // 64) * ((Cc + 63) // 64))
    tab_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(device)
    starts_dev = torch.tensor(starts, dtype=torch.int32).to(device)
    return tab_dev, starts_dev, starts[-1]


def mt_cast_transpose(dtype, tab_dev, starts_dev, n_tensors, n_tiles):
    call('vtx_mt_cast_transpose', _DT[dtype], ptr(tab_dev), ptr(starts_dev), int(n_tensors), int(n_tiles), stream())


def cast_from_f32(src, dtype):
    need_cuda(src)
    if dtype == torch.float32:
        return src
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    call('vtx_cast_from_f32', _DT[dtype], src.numel(), ptr(src.contiguous()), ptr(dst), stream())
    return dst


def cast_to_f32(src):
    need_cuda(src)
    if src.dtype == torch.float32:
        return src
    dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    call('vtx_cast_to_f32', dt(src), src.numel(), ptr(src.contiguous()), ptr(dst), stream())
    return dst


# ---- small host -> device uploads without the copy engine --------------------------------------
# hipMemcpyAsync of tiny buffers interleaved with kernels stalls this stack badly (measured: 33
# DropPath mask uploads per step turned a 43 ms step into ~100 ms with 90 ms host stalls).  Instead
# the values are written into pinned host memory and a copy KERNEL on the compute stream reads them
# over the fabric.  Pinned slots are recycled only after the event recorded behind their kernel.
_pinned_slots = []          # [pinned tensor, event or None]


def upload_f32(values_cpu, device):
    """float32 CPU tensor (any shape) -> 1-D device tensor, stream-ordered, no memcpy call."""
    n = values_cpu.numel()
    slot = None
    for sl in _pinned_slots:
        if sl[0].numel() >= n and (sl[1] is None or sl[1].query()):
            slot = sl
            break
    if slot is None:
        cap = max(1024, 1 << (n - 1).bit_length())
        slot = [torch.empty(cap, dtype=torch.float32, pin_memory=True), None]
        _pinned_slots.append(slot)
    slot[0].numpy()[:n] = values_cpu.reshape(-1).numpy()        # NumPy copy: keep ATen's CPU thread pool out of the step
    out = torch.empty(n, dtype=torch.float32, device=device)
    call('vtx_cast_from_f32', _lib.VTX_F32, n, slot[0].data_ptr(), out.data_ptr(), stream())
    ev = torch.cuda.Event()
    ev.record()
    slot[1] = ev
    return out


_input_norm = None          # (mean[3], std[3]) applied to uint8 clips (set_input_normalization)


def set_input_normalization(mean, std):
    """Mean / std of the reference's transforms.Normalize for uint8 [B,T,H,W,3] clips (the values the
    reference is configured with, e.g. model_pretrain.py's --mean/--std); None disables uint8 input."""
    global _input_norm
    if mean is None:
        _input_norm = None
        return
    mean, std = [float(v) for v in mean], [float(v) for v in std]
    if len(mean) != 3 or len(std) != 3 or any(v == 0.0 for v in std):
        raise ValueError('set_input_normalization: need 3 means and 3 non-zero stds')
    _input_norm = (mean, std)


def clip_dims(clip):
    """(B, T, C, H, W) of a float [B,T,C,H,W] or a uint8 channels-last [B,T,H,W,3] clip."""
    if clip.dtype == torch.uint8:
        B, T, H, W, Cc = clip.shape
        if Cc != 3:
            raise ValueError('uint8 clips must be [B,T,H,W,3]')
        return B, T, Cc, H, W
    B, T, Cc, H, W = clip.shape
    return B, T, Cc, H, W


def patch_rows(clip, dtype, ps, ts, frame_major):
    """[B,T,C,H,W] fp32 -> [B*(T/ts)*P, C*ts*ps*ps] rows in `dtype`; a uint8 [B,T,H,W,3] clip is
    normalised on the fly (ToTensor + Normalize of the reference, see set_input_normalization)."""
    need_cuda(clip)
    if clip.dtype == torch.uint8:
        if _input_norm is None:
            raise ValueError('uint8 clip: call vtx.set_input_normalization(mean, std) first')
        clip = clip.contiguous()
        B, T, Cc, H, W = clip_dims(clip)
        K = Cc * ts * ps * ps
        rows = torch.empty(B * (T // ts) * (H // ps) * (W // ps), K, dtype=dtype, device=clip.device)
        mean = (C.c_float * 3)(*_input_norm[0])
        std = (C.c_float * 3)(*_input_norm[1])
        with _timed('patch_rows', nbytes=clip.numel() + rows.numel() * rows.element_size(), key=f'u8 {B}x{T}'):
            call('vtx_patch_rows_u8', _DT[dtype], B, T, H, W, ps, ts, ptr(clip), mean, std, ptr(rows), K, int(frame_major),
                 stream())
        return rows
    if clip.dtype != torch.float32:
        clip = clip.float()
    clip = clip.contiguous()
    B, T, Cc, H, W = clip.shape
    K = Cc * ts * ps * ps
    rows = torch.empty(B * (T // ts) * (H // ps) * (W // ps), K, dtype=dtype, device=clip.device)
    with _timed('patch_rows', nbytes=clip.numel() * 4 + rows.numel() * rows.element_size(), key=f'f32 {B}x{T}'):
        call('vtx_patch_rows', _DT[dtype], B, T, Cc, H, W, ps, ts, ptr(clip), ptr(rows), K, int(frame_major), stream())
    return rows


def embed_table(dtype, P, T, D, bias, pos, time_embed, cls, frame_major=False):
    dev = pos.device
    E = torch.empty(P * T, D, dtype=dtype, device=dev)
    cls_row = torch.empty(D, dtype=dtype, device=dev) if cls is not None else None
    call('vtx_embed_table', _DT[dtype], P, T, D, ptr(bias), ptr(pos), ptr(time_embed), ptr(cls), ptr(E),
         ptr(cls_row), int(frame_major), stream())
    return E, cls_row


# ------------------------------------------------------------------------ HOG
_hog_tables = {}


def hog_table(device):
    key = str(device)
    if key not in _hog_tables:
        host = torch.empty(_lib.load().vtx_hog_table_bytes() // 8, dtype=torch.float64)     # magnitudes + the correction words
        call('vtx_hog_build_table', host.data_ptr())
        _hog_tables[key] = host.to(device)
    return _hog_tables[key]


def hog_fwd(frames, want_bins=False):
    """frames uint8 [F,H,W,3] (CUDA) -> float64 [F,H/16,W/16,108] (+ int32 bins [F,3,H,W])."""
    need_cuda(frames)
    if frames.dtype != torch.uint8 or frames.ndim != 4 or frames.shape[-1] != 3:
        raise TypeError('hog_fwd: frames must be uint8 [F,H,W,3]')
    frames = frames.contiguous()
    F, H, W, _ = frames.shape
    out = torch.empty(F, H // 16, W // 16, 108, dtype=torch.float64, device=frames.device)
    bins = torch.empty(F, 3, H, W, dtype=torch.int32, device=frames.device) if want_bins else None
    table = hog_table(frames.device)
    with _timed('hog', nbytes=frames.numel() + out.numel() * 8, key=f'{F}x{H}x{W}'):
        call('vtx_hog_fwd', ptr(frames), F, H, W, ptr(table), table.numel() * 8, ptr(out), ptr(bins), stream())
    return (out, bins) if want_bins else out


# ------------------------------------------------- clip-batch mixing, accuracy (csrc/head.hip)
def mixup_batch_(x, lam):
    """In place: x[b] = x[b]*lam + x[B-1-b]*(1-lam) on a contiguous fp32 [B, ...] batch (reference mixup.py:112-113)."""
    need_cuda(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise TypeError('mixup_batch_: contiguous float32 batch required')
    import numpy as np
    B = x.shape[0]
    # ATen multiplies a float32 tensor by a Python scalar after casting the scalar to float32
    call('vtx_mixup_batch', ptr(x), B, x.numel() // B, float(np.float32(lam)), float(np.float32(1. - lam)), stream())
    return x


def cutmix_batch_(x, yl, yh, xl, xh):
    """In place: x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh] on a contiguous fp32 [B, planes, H, W] batch."""
    need_cuda(x)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.ndim != 4:
        raise TypeError('cutmix_batch_: contiguous float32 [B, planes, H, W] batch required')
    B, Pn, H, W = x.shape
    call('vtx_cutmix_batch', ptr(x), B, Pn, H, W, int(yl), int(yh), int(xl), int(xh), stream())
    return x


def mixup_target(labels, num_classes, lam, smoothing):
    """Label-smoothed one-hot rows mixed with the flipped batch's (reference mixup.py:20-25) -> fp32 [B, num_classes]."""
    import numpy as np
    need_cuda(labels)
    labels = labels.long().contiguous().view(-1)
    B = labels.numel()
    off = smoothing / num_classes
    on = 1. - smoothing + off
    out = torch.empty(B, num_classes, dtype=torch.float32, device=labels.device)
    call('vtx_mixup_target', ptr(labels), B, int(num_classes), float(np.float32(on)), float(np.float32(off)),
         float(np.float32(lam)), float(np.float32(1. - lam)), ptr(out), stream())
    return out


def topk_correct(scores, labels, k, counter):
    """counter (int32 device scalar) += rows of fp32 ``scores`` [B,C] whose label ranks among the k largest."""
    need_cuda(scores, labels, counter)
    scores = scores.float().contiguous()
    labels = labels.long().contiguous()
    B, Cn = scores.shape
    call('vtx_topk_correct', ptr(scores), ptr(labels), B, Cn, int(k), ptr(counter), stream())


def gelu_grad_mul(dy, h, out):
    """out = dy * gelu'(h) elementwise (same shapes / dtype, numel % 8 == 0)."""
    need_cuda(dy, h, out)
    call('vtx_gelu_grad_mul', dt(dy), dy.numel(), ptr(dy), ptr(h), ptr(out), stream())
