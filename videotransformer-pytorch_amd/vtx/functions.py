"""torch.autograd.Function wrappers: one per reference sub-module forward, each
running its whole forward / backward as a short chain of libvtx kernels.

The residual stream x is [B, 1 + P*T, D] in the compute dtype (float32 = exact
path, bfloat16 = MFMA bf16 path), token index = 1 + p*T + t (reference
video_transformer.py:228,236).  Parameters stay float32; their compute-dtype
copies (and the transposes used by the input-gradient GEMMs) are staged by
``weights()`` and cached per parameter version.
"""
import weakref

import torch

from . import ops
from ._lib import IDENT, ATTN_CONTIG, ATTN_SPACE

_wcache = {}


def weights(p, dtype, need_t=None):
    """(W, W^T) copies of Linear weight ``p`` [out,in] in ``dtype``; cached on the parameter OBJECT
    (weak reference).  A cached copy is reused only while the parameter's version counter, storage
    address and device are unchanged: in-place updates through ``p`` (optimizers, ``copy_`` under
    no_grad) bump the version; ``p.data = ...``, ``module.to(device)`` and re-allocation change the
    address or device.  An update written through ``p.data`` IN PLACE (``p.data.copy_(...)``, EMA
    loops) changes none of the three -- such code must call ``clear_weight_cache()`` afterwards
    (``vtx.dp.broadcast_parameters`` does).  Forward passes fetch both copies and hand W^T to backward
    through ctx.save_for_backward (ctx.saved_tensors returns new tensor objects, and storage addresses
    are recycled between models, so neither id() nor data_ptr() of a saved tensor alone is a safe key)."""
    if need_t is None:          # NB: grad mode is off inside autograd.Function.forward -- callers there pass
        need_t = torch.is_grad_enabled()   # any(ctx.needs_input_grad) explicitly
    key = (id(p), dtype)
    stamp = (p._version, p.data_ptr(), p.device)
    hit = _wcache.get(key)
    if hit is not None and hit[0]() is p and hit[1] == stamp and (hit[3] is not None or not need_t):
        return hit[2], hit[3]
    w2d = p.detach().reshape(p.shape[0], -1)
    wc, wt = ops.cast_transpose(w2d, dtype, want_c=True, want_t=need_t)
    if len(_wcache) > 4096:                       # dead parameters: drop stale entries
        for k in [k for k, v in _wcache.items() if v[0]() is None]:
            del _wcache[k]
    _wcache[key] = (weakref.ref(p), stamp, wc, wt)
    return wc, wt


_restage_tables = {}


def restage_weights(params):
    """Refresh, in ONE launch per compute dtype, the staged W / W^T copies of every parameter in ``params`` that has
    them -- for callers that have just updated those parameters in place (the fused optimizers do this at the end of
    step(): the per-parameter refresh in ``weights()`` is ~86 launches of ~13 us per TimeSformer-B step).  The copies
    are overwritten in place: no autograd graph that saved them may still be waiting for its backward."""
    groups = {}
    for p in params:
        for dtype in (torch.bfloat16, torch.float32):
            hit = _wcache.get((id(p), dtype))
            if hit is None or hit[0]() is not p or not p.is_cuda:
                continue
            wc, wt = hit[2], hit[3]
            if dtype == torch.float32:
                wc = None                                  # the fp32 "copy" is the parameter itself
            if wc is None and wt is None:
                continue
            if (p._version, p.data_ptr(), p.device) == hit[1]:
                continue                                   # not modified since it was staged
            groups.setdefault((dtype, p.device), []).append((p, hit, wc, wt))
    for (dtype, dev), ents in groups.items():
        key = (dtype, dev, tuple((p.data_ptr(), 0 if wc is None else wc.data_ptr(), 0 if wt is None else wt.data_ptr())
                                 for p, _, wc, wt in ents))
        tabs = _restage_tables.get((dtype, dev))
        if tabs is None or tabs[0] != key:
            tab_dev, starts_dev, n_tiles = ops.ct_table([(p.detach().reshape(p.shape[0], -1), wc, wt) for p, _, wc, wt in ents], dev)
            tabs = (key, tab_dev, starts_dev, n_tiles)
            _restage_tables[(dtype, dev)] = tabs
        ops.mt_cast_transpose(dtype, tabs[1], tabs[2], len(ents), tabs[3])
        for p, hit, wc, wt in ents:
            _wcache[(id(p), dtype)] = (hit[0], (p._version, p.data_ptr(), p.device), hit[2], hit[3])
            # the copies were rewritten behind autograd's back: bump their version counters (an in-place no-op on a
            # zero-element view, no launch), so that a graph that saved them BEFORE this update -- optimizer.step() between
            # a forward and its backward -- fails with autograd's "modified by an inplace operation" error, not silently
            for t in (wc, wt):
                if t is not None:
                    t.view(-1)[:0].zero_()


_mcache = {}
_merge_tfc = True


def set_merge_temporal_fc(on):
    """TimeAttnFn: run attn.proj and temporal_fc as ONE GEMM with the product weight (default) or as the two GEMMs
    of the reference (transformer.py:268-275)."""
    global _merge_tfc
    _merge_tfc = bool(on)


def merged_proj(proj_w, proj_b, tfc_w, tfc_b, dtype, need_t, c=1.0):
    """attn.proj followed by temporal_fc is linear in the attention output (only a per-sequence DropPath scale sits
    between them): W_c = W_tfc W_proj and the epilogue bias b_c + b_tfc / c with b_c = W_tfc b_proj, in fp32 from the
    fp32 parameters by ONE vtx_wprod launch (exact-fp32 matrix instruction; weights x weights, no activation passes
    through it), staged like any other weight.  Returns (W_c, W_c^T, bias); cached while the four parameters and the
    keep scale c are unchanged (same validity rule as ``weights``)."""
    ps = (proj_w, proj_b, tfc_w, tfc_b)
    key = (id(proj_w), id(tfc_w), dtype)
    stamp = tuple((p._version, p.data_ptr(), p.device) for p in ps) + (float(c),)
    hit = _mcache.get(key)
    if hit is not None and all(r() is p for r, p in zip(hit[0], ps)) and hit[1] == stamp and (hit[3] is not None or not need_t):
        return hit[2], hit[3], hit[4]
    wc32, bias = ops.wprod(tfc_w.detach(), proj_w.detach(), x=proj_b.detach(), z=tfc_b.detach(), beta_z=1.0 / float(c))
    wc, wt = ops.cast_transpose(wc32, dtype, want_c=True, want_t=need_t)
    if len(_mcache) > 1024:
        for k in [k for k, v in _mcache.items() if any(r() is None for r in v[0])]:
            del _mcache[k]
    _mcache[key] = (tuple(weakref.ref(p) for p in ps), stamp, wc, wt, bias)
    return wc, wt, bias


def clear_weight_cache():
    """Drop every staged compute-dtype weight copy (call after out-of-band parameter updates that the cache
    validity check cannot see)."""
    _wcache.clear()
    _mcache.clear()
    _restage_tables.clear()


# ---- direct parameter gradients ---------------------------------------------------------------------
# autograd accumulates a returned parameter gradient with one `grad += new` kernel per parameter (and the
# Functions below used to zero-fill the LayerNorm gradients they accumulate into): ~370 five-microsecond launches
# per TimeSformer-B step.  With direct gradients on, the weight-gradient GEMM's reduction, its fused bias
# column sums and the LayerNorm backward accumulate straight into ``param.grad`` (same fp32 `+=`), the Function
# returns None for that parameter and calls the parameter's post-accumulate-grad hooks itself, right behind the
# kernel launch (the data-parallel bucket hooks of vtx.dp; torch 2.10 runs those hooks once more when it sees the
# None, so hooks must count a parameter once per backward -- GradBuckets does).  Opt-in (vtx.dp.GradBuckets(..., direct=True) turns it on): it needs
# pre-allocated contiguous fp32 ``.grad`` buffers, and only ``loss.backward()`` sees these gradients --
# ``torch.autograd.grad`` and tensor hooks on the parameters do not.
_direct = False


def set_direct_grads(on):
    global _direct
    _direct = bool(on)


def direct_grads_enabled():
    return _direct


def _sink(p):
    if not _direct or p is None:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_cuda or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


_firing = False      # True while _fire() runs the hooks: lets a hook tell the kernels' own call from autograd's


def _fire(*params):
    global _firing
    _firing = True
    try:
        for p in params:
            hooks = getattr(p, '_post_accumulate_grad_hooks', None)
            if hooks:
                for h in list(hooks.values()):
                    h(p)
    finally:
        _firing = False


def firing():
    """True inside a hook call made by the direct-gradient kernels themselves (vtx.dp.GradBuckets uses it to tell a second
    accumulation into an already counted parameter -- a module applied twice in one backward -- from autograd's own
    duplicate hook call for the None the Function returned)."""
    return _firing


def _linear_grads(w, b, A, Bm, M, N1, N2, **kw):
    """(d_weight, d_bias) of a Linear out of the two GEMM operands; (None, None) when they went straight into
    w.grad / b.grad."""
    gw, gb = _sink(w), _sink(b)
    if gw is not None and gb is not None:
        ops.gemm_tn(A, Bm, M, N1, N2, out=gw.view(N1, N2), accumulate=True, colsum_out=gb, colsum_accumulate=True, **kw)
        _fire(w, b)
        return None, None
    return ops.gemm_tn(A, Bm, M, N1, N2, want_colsum=True, **kw)


def _ln_grad_buffers(ln_w, ln_b, D, device):
    """Buffers vtx_layernorm_bwd accumulates d_gamma / d_beta into: the parameters' own .grad (direct), else zeros."""
    gw, gb = _sink(ln_w), _sink(ln_b)
    if gw is not None and gb is not None:
        return gw, gb, True
    return (torch.zeros(D, dtype=torch.float32, device=device), torch.zeros(D, dtype=torch.float32, device=device), False)


# ---- the exact residual stream -------------------------------------------------------------------------
# vtx.set_stream('fp32'): a sub-block returns its CONTRIBUTION d = f(x) (bf16) next to the float32 stream it read, and the next
# sub-block's LayerNorm kernel forms x = xs + d in float32 (vtx_layernorm_acc_fwd) -- no bf16 rounding of the running sum.  The
# float32 stream is a side buffer, not an autograd tensor: gradients flow through the chain of contributions exactly as they
# flow through the bf16 stream (d(xs + d)/dd = 1), so every backward below is shared between the two modes.  Every attention
# type of TimeSformer / ViViT (readers that need the stream as ONE tensor -- the space_only frame mean, the ViViT fact-encoder
# glue -- take it through StreamValueFn); MViT keeps its bf16 stream.
_exact = False


def set_exact_stream(on):
    global _exact
    _exact = bool(on)


def exact_stream():
    return _exact


_exact_grad = False


def set_exact_grad_stream(on):
    """With the exact residual stream: keep the stream's GRADIENT in float32 too (vtx_layernorm_bwd_g32).  The backward of a
    sub-block then reads the float32 gradient of its output next to the bf16 rounding autograd hands it, adds its LayerNorm
    backward in float32 and hands both on: the running sum of the branch gradients is never rounded to bf16 -- as under
    torch.autocast, where the stream and therefore its gradient are float32 and only the branches compute in bf16."""
    global _exact_grad
    _exact_grad = bool(on)


def exact_grad_stream():
    return _exact_grad


def _grad_stream(dout, x_stream):
    """The float32 stream gradient `dout` is the bf16 rounding of: the buffer the producing backward attached to it (valid
    only while `dout` is that very tensor, unmodified: autograd's in-place accumulation of a second consumer's gradient bumps
    the version counter, an out-of-place sum or a view is another object), else a float32 copy of `dout` -- the gradient
    stream starts (or restarts) there.  None when the mode is off or the block did not run under the exact stream."""
    if not _exact_grad or x_stream.dtype != torch.float32 or dout.dtype != torch.bfloat16:
        return None
    h = getattr(dout, '_vtx_g32', None)
    if h is not None and h[1] == dout._version and h[0].shape == dout.shape and h[0].device == dout.device:
        return h[0]
    return ops.cast_to_f32(dout)


def _hand_on(dx, dx32):
    if dx32 is not None:
        dx._vtx_g32 = (dx32, dx._version)
    return dx


def _ln_bwd_res(dxn, x_stream, xmap, rows, D, mean, rstd, ln_w, dout, dx, d_ln_w, d_ln_b, g32, dx32):
    """dx = dout + LayerNorm-backward(dxn) on the mapped rows; with g32 / dx32 in float32 (dx its bf16 rounding)."""
    if g32 is not None:
        ops.layernorm_bwd(dxn, D, IDENT, x_stream, D, xmap, rows, D, mean, rstd, ln_w, None, dx, D, d_ln_w, d_ln_b, dres32=g32, dx32=dx32)
    else:
        ops.layernorm_bwd(dxn, D, IDENT, x_stream, D, xmap, rows, D, mean, rstd, ln_w, dout, dx, D, d_ln_w, d_ln_b)


def _copy_rows_res(dout, dx, g32, dx32, rows, D, rowmap):
    """Rows the block does not touch: their gradient passes through (both forms of it)."""
    ops.row_scale_copy(dout, dx, rows, D, smap=rowmap, dmap=rowmap)
    if g32 is not None:
        ops.row_scale_copy(g32, dx32, rows, D, smap=rowmap, dmap=rowmap)


_zero1 = {}


def _zero_rows(finite_src, dst, rows, D, rowmap):
    """dst[rowmap(m)] = 0 for m < rows (0 * finite_src[rowmap(m)]: the source only has to be finite)."""
    z = _zero1.get(dst.device)
    if z is None:
        z = _zero1[dst.device] = torch.zeros(1, dtype=torch.float32, device=dst.device)
    ops.row_scale_copy(finite_src, dst, rows, D, smap=rowmap, dmap=rowmap, s=z, rs=(1, 0, 1, 0))


def _empty(shape, like, dtype=None):
    return torch.empty(shape, dtype=dtype or like.dtype, device=like.device)


def _chk(x):
    ops.need_cuda(x)
    if not x.is_contiguous():
        x = x.contiguous()
    return x


# ---------------------------------------------------------------------------------
class TimeAttnFn(torch.autograd.Function):
    """DividedTemporalAttentionWithPreNorm.forward, use_cls_token=False (reference transformer.py:234-282).

    attn.proj and temporal_fc have only the per-sequence DropPath scale s between them, so
        out = x + b_tfc + s * (o W_c^T + b_c),      W_c = W_tfc W_proj,  b_c = W_tfc b_proj
    is ONE GEMM (``merged_proj``; three launches of 150k x 768 x 768 less per layer and step than the two Linear
    layers: forward, input gradient, weight gradient).  With s in {0, c}: the epilogue computes
    s * (acc + b_c + b_tfc / c) + x, and the rows of dropped sequences (s = 0) are set to x + b_tfc by
    vtx_dropped_rows_fix, which also zeroes their rows of o -- they then drop out of the weight-gradient product, and
    the attention backward never needs them (their do is 0).  Backward, with G = c * dout^T o (kept rows):
        dW_tfc = G W_proj^T + u b_proj^T,  dW_proj = W_tfc^T G,  db_proj = W_tfc^T u,  db_tfc = colsum(dout),
        u = c * (colsum(dout) - colsum over dropped rows of dout).
    ``set_merge_temporal_fc(False)`` runs the two GEMMs of the reference instead."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, tfc_w, tfc_b, T, heads, scale_vec, eps=1e-5,
                keep_scale=None, xs=None, exact=False):
        # exact: x is the previous sub-block's contribution d, xs the float32 stream it was computed from (None: the stream
        # starts at d); returns (this block's contribution, the float32 stream xs + d)
        x = _chk(x)
        B, N1, D = x.shape
        N = N1 - 1
        M = B * N
        hd = D // heads
        tm = ops.tokmap(N)
        dtp = x.dtype
        need_t = any(ctx.needs_input_grad)
        merged = _merge_tfc and (scale_vec is None or keep_scale is not None) and proj_b is not None and tfc_b is not None
        xn = _empty((M, D), x)
        mean = _empty((M,), x, torch.float32)
        rstd = _empty((M,), x, torch.float32)
        x32 = None
        if exact:
            x32 = torch.empty(B, N1, D, dtype=torch.float32, device=x.device)
            ops.layernorm_acc_fwd(xs, x, M, D, D, tm, x32, D, tm, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
            ops.layernorm_acc_fwd(xs, x, B, D, D, ops.clsmap(N), x32, D, ops.clsmap(N))       # the cls rows: accumulate only
        else:
            ops.layernorm_fwd(x, M, D, D, tm, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
        res = None if exact else x                  # the residual the GEMM epilogue adds: none under the exact stream
        wq, wqT = weights(qkv_w, dtp, need_t)
        qkv = _empty((M, 3 * D), x)
        ops.gemm_nt(xn, wq, qkv, M, 3 * D, D, bias=qkv_b)
        o = _empty((M, D), x)
        S = M // T
        lse = _empty((S * heads * T,), x, torch.float32)
        scale = hd ** -0.5
        ops.attn_fwd(qkv, o, lse, ATTN_CONTIG, S, T, heads, hd, scale)
        out = torch.empty_like(x)
        if merged:
            c = 1.0 if scale_vec is None else float(keep_scale)
            wc, wcT, bias = merged_proj(proj_w, proj_b, tfc_w, tfc_b, dtp, need_t, c)
            ops.gemm_nt(o, wc, out, M, D, D, cmap=tm, bias=bias, R=res, rmap=tm, row_scale=scale_vec, rs=(T, 1, 1, 0))
            if scale_vec is not None:
                ops.dropped_rows_fix(scale_vec, M, D, T, x=res, xmap=tm, bias=tfc_b.detach(), out=out, omap=tm, zero=o)
            a = x.new_empty(0)
            wts = [t for t in (wqT, wcT) if t is not None]
        else:
            wp, wpT = weights(proj_w, dtp, need_t)
            a = _empty((M, D), x)
            ops.gemm_nt(o, wp, a, M, D, D, bias=proj_b, row_scale=scale_vec, rs=(T, 1, 1, 0))
            wt, wtT = weights(tfc_w, dtp, need_t)
            ops.gemm_nt(a, wt, out, M, D, D, cmap=tm, bias=tfc_b, R=res, rmap=tm)
            wts = [t for t in (wqT, wpT, wtT) if t is not None]
        if exact:
            _zero_rows(x, out, B, D, ops.clsmap(N))                 # the block contributes nothing to the cls rows
        else:
            ops.row_scale_copy(x, out, B, D, smap=ops.clsmap(N), dmap=ops.clsmap(N))
        ctx.save_for_backward(x32 if exact else x, ln_w, mean, rstd, xn, qkv, o, lse, a,
                              scale_vec if scale_vec is not None else x.new_empty(0), *wts)
        ctx.cfg = (T, heads, scale_vec is not None, merged, keep_scale)
        ctx.params = (ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, tfc_w, tfc_b)
        if exact:
            ctx.mark_non_differentiable(x32)
            ctx.set_materialize_grads(False)         # no zero-filled 'gradient' of the float32 stream (0.46 GB per sub-block at 96 clips)
            return out, x32
        return out

    @staticmethod
    def backward(ctx, dout, _dstream=None):
        # (x: the saved stream -- bf16, or float32 under the exact stream; every buffer below takes its dtype from dout)
        x, ln_w, mean, rstd, xn, qkv, o, lse, a, sv, wqT, *wrest = ctx.saved_tensors
        p_ln_w, p_ln_b, p_qkv_w, p_qkv_b, p_proj_w, p_proj_b, p_tfc_w, p_tfc_b = ctx.params
        T, heads, has_scale, merged, keep_scale = ctx.cfg
        sv = sv if has_scale else None
        dout = _chk(dout)
        B, N1, D = x.shape
        N = N1 - 1
        M = B * N
        hd = D // heads
        S = M // T
        tm = ops.tokmap(N)
        dtp = dout.dtype
        x_stream, x = x, dout                          # allocation template from here on
        do = _empty((M, D), x)
        if merged:
            wcT, = wrest
            c = float(keep_scale) if has_scale else 1.0
            G, cs_all = ops.gemm_tn(dout, o, M, D, D, amap=tm, want_colsum=True)
            ops.gemm_nt(dout, wcT, do, M, D, D, amap=tm, row_scale=sv, rs=(T, 1, 1, 0))
            if has_scale:                              # u = c * (colsum(dout) - colsum over the dropped rows of dout)
                part = ops.dropped_rows_colsum(dout, sv, M, D, T, smap=tm)
                u = ops.reduce_rows(cs_all, 1, 1, D, D, 0, 1, 0, scale=c)
                ops.reduce_rows(part, 1, part.shape[0], D, D, 0, 1, 0, out=u, scale=-c, accumulate=True)
                u = u.view(-1)
            else:
                u = cs_all
            # the merged weight gradient G mapped back to the two Linear layers (vtx_wprod: fp32, weights x weights):
            #   dW_tfc = c G W_proj^T + u b_proj^T,   dW_proj = c W_tfc^T G,   db_proj = W_tfc^T u,   db_tfc = colsum(dout)
            pw, pb, tw = p_proj_w.detach(), p_proj_b.detach(), p_tfc_w.detach()
            sinks = [_sink(p) for p in (p_tfc_w, p_tfc_b, p_proj_w, p_proj_b)]
            if all(g is not None for g in sinks):      # the same `grad += new` autograd would do, inside the kernels
                ops.wprod(G, pw, tb=True, alpha=c, u=u, v=pb, out=sinks[0], accumulate=True)
                ops.wprod(tw, G, ta=True, alpha=c, out=sinks[2], accumulate=True, x=u, y=sinks[3], y_accumulate=True)
                ops.reduce_rows(cs_all, 1, 1, D, D, 0, 1, 0, out=sinks[1].view(1, D), accumulate=True)
                _fire(p_tfc_w, p_tfc_b, p_proj_w, p_proj_b)
                d_tfc_w = d_tfc_b = d_proj_w = d_proj_b = None
            else:
                d_tfc_w = ops.wprod(G, pw, tb=True, alpha=c, u=u, v=pb)
                d_proj_w, d_proj_b = ops.wprod(tw, G, ta=True, alpha=c, x=u)
                d_tfc_b = cs_all
        else:
            wpT, wtT = wrest
            # temporal_fc
            d_tfc_w, d_tfc_b = _linear_grads(p_tfc_w, p_tfc_b, dout, a, M, D, D, amap=tm)
            da = _empty((M, D), x)
            ops.gemm_nt(dout, wtT, da, M, D, D, amap=tm, row_scale=sv, rs=(T, 1, 1, 0))
            # proj
            d_proj_w, d_proj_b = _linear_grads(p_proj_w, p_proj_b, da, o, M, D, D)
            ops.gemm_nt(da, wpT, do, M, D, D)
        # attention core
        dqkv = _empty((M, 3 * D), x)
        ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_CONTIG, S, T, heads, hd, hd ** -0.5)
        d_qkv_w, d_qkv_b = _linear_grads(p_qkv_w, p_qkv_b, dqkv, xn, M, 3 * D, D)
        dxn = _empty((M, D), x)
        ops.gemm_nt(dqkv, wqT, dxn, M, D, 3 * D)
        # LayerNorm + residual
        dx = torch.empty_like(x)
        g32 = _grad_stream(dout, x_stream)
        dx32 = torch.empty_like(g32) if g32 is not None else None
        d_ln_w, d_ln_b, direct = _ln_grad_buffers(p_ln_w, p_ln_b, D, x.device)
        _ln_bwd_res(dxn, x_stream, tm, M, D, mean, rstd, ln_w, dout, dx, d_ln_w, d_ln_b, g32, dx32)
        if direct:
            _fire(p_ln_w, p_ln_b)
            d_ln_w = d_ln_b = None
        _copy_rows_res(dout, dx, g32, dx32, B, D, ops.clsmap(N))
        _hand_on(dx, dx32)
        return (dx, d_ln_w, d_ln_b, d_qkv_w, d_qkv_b, d_proj_w, d_proj_b, d_tfc_w, d_tfc_b, None, None, None, None, None, None, None)


# ---------------------------------------------------------------------------------
class SpaceAttnFn(torch.autograd.Function):
    """DividedSpatialAttentionWithPreNorm.forward, use_cls_token=True
    (reference transformer.py:336-382).  LayerNorm and the qkv / proj Linears are
    row-wise, so they run once over the natural token order (the cls row once per
    clip instead of once per frame); only the attention kernel regroups rows."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, T, heads, scale_vec, want_probs, eps=1e-5, xs=None, exact=False):
        x = _chk(x)                                  # exact: the contribution d of the previous sub-block (see TimeAttnFn)
        B, N1, D = x.shape
        N = N1 - 1
        P = N // T
        M1 = B * N1
        hd = D // heads
        dtp = x.dtype
        xn = _empty((M1, D), x)
        mean = _empty((M1,), x, torch.float32)
        rstd = _empty((M1,), x, torch.float32)
        x32 = None
        if exact:
            x32 = torch.empty(B, N1, D, dtype=torch.float32, device=x.device)
            ops.layernorm_acc_fwd(xs, x, M1, D, D, IDENT, x32, D, IDENT, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
        else:
            ops.layernorm_fwd(x, M1, D, D, IDENT, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
        wq, wqT = weights(qkv_w, dtp, any(ctx.needs_input_grad))
        qkv = _empty((M1, 3 * D), x)
        ops.gemm_nt(xn, wq, qkv, M1, 3 * D, D, bias=qkv_b)
        Mo = B * N + B * T
        o = _empty((Mo, D), x)
        S, L = B * T, P + 1
        lse = _empty((S * heads * L,), x, torch.float32)
        probs = _empty((S, heads, L, L), x, torch.float32) if want_probs else None
        ops.attn_fwd(qkv, o, lse, ATTN_SPACE, S, L, heads, hd, hd ** -0.5, B, T, P, probs=probs)
        if want_probs:
            ctx.mark_non_differentiable(probs)
            return probs
        wp, wpT = weights(proj_w, dtp, any(ctx.needs_input_grad))
        out = torch.empty_like(x)
        a_cls = _empty((B * T, D), x)
        tm = ops.tokmap(N)
        ops.gemm_nt(o, wp, out, Mo, D, D, cmap=tm, bias=proj_b, row_scale=scale_vec, rs=(N, T, T, 1),
                    R=None if exact else x, rmap=tm, split_row=B * N, Csplit=a_cls)
        ops.cls_mean_fwd(a_cls, None if exact else x, out, B, T, D, N1)
        ctx.save_for_backward(x32 if exact else x, ln_w, mean, rstd, xn, qkv, o, lse,
                              scale_vec if scale_vec is not None else x.new_empty(0),
                              *[t for t in (wqT, wpT) if t is not None])
        ctx.cfg = (T, heads, scale_vec is not None)
        ctx.params = (ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b)
        if exact:
            ctx.mark_non_differentiable(x32)
            ctx.set_materialize_grads(False)         # no zero-filled 'gradient' of the float32 stream (0.46 GB per sub-block at 96 clips)
            return out, x32
        return out

    @staticmethod
    def backward(ctx, dout, _dstream=None):
        x, ln_w, mean, rstd, xn, qkv, o, lse, sv, wqT, wpT = ctx.saved_tensors
        p_ln_w, p_ln_b, p_qkv_w, p_qkv_b, p_proj_w, p_proj_b = ctx.params
        T, heads, has_scale = ctx.cfg
        sv = sv if has_scale else None
        dout = _chk(dout)
        B, N1, D = x.shape
        N = N1 - 1
        P = N // T
        M1 = B * N1
        Mo = B * N + B * T
        hd = D // heads
        dtp = dout.dtype
        x_stream, x = x, dout                          # (the saved stream may be float32: buffers take dout's dtype)
        da = _empty((Mo, D), x)
        ops.space_grad_prep(dout, sv, da, B, T, P, D)
        d_proj_w, d_proj_b = _linear_grads(p_proj_w, p_proj_b, da, o, Mo, D, D)
        do = _empty((Mo, D), x)
        ops.gemm_nt(da, wpT, do, Mo, D, D)
        dqkv = _empty((M1, 3 * D), x)
        dqkv_cls = _empty((B * T, 3 * D), x)
        ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_SPACE, B * T, P + 1, heads, hd, hd ** -0.5, B, T, P,
                     dqkv_cls=dqkv_cls)
        ops.cls_qkv_reduce(dqkv_cls, dqkv, B, T, 3 * D, N1)
        d_qkv_w, d_qkv_b = _linear_grads(p_qkv_w, p_qkv_b, dqkv, xn, M1, 3 * D, D)
        dxn = _empty((M1, D), x)
        ops.gemm_nt(dqkv, wqT, dxn, M1, D, 3 * D)
        dx = torch.empty_like(x)
        g32 = _grad_stream(dout, x_stream)
        dx32 = torch.empty_like(g32) if g32 is not None else None
        d_ln_w, d_ln_b, direct = _ln_grad_buffers(p_ln_w, p_ln_b, D, x.device)
        _ln_bwd_res(dxn, x_stream, IDENT, M1, D, mean, rstd, ln_w, dout, dx, d_ln_w, d_ln_b, g32, dx32)
        if direct:
            _fire(p_ln_w, p_ln_b)
            d_ln_w = d_ln_b = None
        _hand_on(dx, dx32)
        return (dx, d_ln_w, d_ln_b, d_qkv_w, d_qkv_b, d_proj_w, d_proj_b, None, None, None, None, None, None, None)


class SelfAttnFn(torch.autograd.Function):
    """MultiheadAttentionWithPreNorm.forward (reference transformer.py:428-456):
    x [Bn, L, D] -> x + DropPath(proj(attn(LN(x))))."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, heads, scale_vec, want_probs, eps=1e-5, xs=None, exact=False):
        x = _chk(x)                                  # exact: the contribution d of the previous sub-block (see TimeAttnFn)
        Bn, L, D = x.shape
        M = Bn * L
        hd = D // heads
        dtp = x.dtype
        xn = _empty((M, D), x)
        mean = _empty((M,), x, torch.float32)
        rstd = _empty((M,), x, torch.float32)
        x32 = None
        if exact:
            x32 = torch.empty(Bn, L, D, dtype=torch.float32, device=x.device)
            ops.layernorm_acc_fwd(xs, x, M, D, D, IDENT, x32, D, IDENT, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
        else:
            ops.layernorm_fwd(x, M, D, D, IDENT, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
        wq, wqT = weights(qkv_w, dtp, any(ctx.needs_input_grad))
        qkv = _empty((M, 3 * D), x)
        ops.gemm_nt(xn, wq, qkv, M, 3 * D, D, bias=qkv_b)
        o = _empty((M, D), x)
        lse = _empty((Bn * heads * L,), x, torch.float32)
        probs = _empty((Bn, heads, L, L), x, torch.float32) if want_probs else None
        ops.attn_fwd(qkv, o, lse, ATTN_CONTIG, Bn, L, heads, hd, hd ** -0.5, probs=probs)
        if want_probs:
            ctx.mark_non_differentiable(probs)
            return probs
        wp, wpT = weights(proj_w, dtp, any(ctx.needs_input_grad))
        out = torch.empty_like(x)
        ops.gemm_nt(o, wp, out, M, D, D, bias=proj_b, row_scale=scale_vec, rs=(L, 1, 1, 0), R=None if exact else x)
        ctx.save_for_backward(x32 if exact else x, ln_w, mean, rstd, xn, qkv, o, lse,
                              scale_vec if scale_vec is not None else x.new_empty(0),
                              *[t for t in (wqT, wpT) if t is not None])
        ctx.cfg = (heads, scale_vec is not None)
        if exact:
            ctx.mark_non_differentiable(x32)
            ctx.set_materialize_grads(False)         # no zero-filled 'gradient' of the float32 stream (0.46 GB per sub-block at 96 clips)
            return out, x32
        return out

    @staticmethod
    def backward(ctx, dout, _dstream=None):
        x, ln_w, mean, rstd, xn, qkv, o, lse, sv, wqT, wpT = ctx.saved_tensors
        heads, has_scale = ctx.cfg
        dout = _chk(dout)
        Bn, L, D = x.shape
        M = Bn * L
        hd = D // heads
        dtp = dout.dtype
        x_stream, x = x, dout                          # (the saved stream may be float32: buffers take dout's dtype)
        if has_scale:
            da = _empty((M, D), x)
            ops.row_scale_copy(dout, da, M, D, s=sv, rs=(L, 1, 1, 0))
        else:
            da = dout
        d_proj_w, d_proj_b = ops.gemm_tn(da, o, M, D, D, want_colsum=True)
        do = _empty((M, D), x)
        ops.gemm_nt(da, wpT, do, M, D, D)
        dqkv = _empty((M, 3 * D), x)
        ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_CONTIG, Bn, L, heads, hd, hd ** -0.5)
        d_qkv_w, d_qkv_b = ops.gemm_tn(dqkv, xn, M, 3 * D, D, want_colsum=True)
        dxn = _empty((M, D), x)
        ops.gemm_nt(dqkv, wqT, dxn, M, D, 3 * D)
        dx = torch.empty_like(x)
        d_ln_w = torch.zeros(D, dtype=torch.float32, device=x.device)
        d_ln_b = torch.zeros(D, dtype=torch.float32, device=x.device)
        g32 = _grad_stream(dout, x_stream)
        dx32 = torch.empty_like(g32) if g32 is not None else None
        _ln_bwd_res(dxn, x_stream, IDENT, M, D, mean, rstd, ln_w, dout, dx, d_ln_w, d_ln_b, g32, dx32)
        _hand_on(dx, dx32)
        return (dx, d_ln_w, d_ln_b, d_qkv_w, d_qkv_b, d_proj_w, d_proj_b, None, None, None, None, None, None)


_compact = True


def set_compact_droppath(on):
    """FFNFn: skip the rows of the clips DropPath drops (default) or compute every clip and multiply by zero."""
    global _compact
    _compact = bool(on)


def _compaction_plan(scale_vec, n_units, rows_per, device):
    """DropPath at the FFN drops whole clips (reference transformer.py:34-42: one draw per sample, :543 p up to 0.1): their
    LayerNorm, both GEMMs and all of their backward are multiplied by zero.  With the mask's host copy at hand (the draw
    happens on the CPU generator, as in the reference) the block runs on the KEPT clips only: logical row m of the compact
    problem is row m + tab[m // rows_per] of the stream (table row map, include/vtx.h).  -> None when nothing is dropped, else
    (kept count, dropped count, largest step of the keep / drop tables, device buffer [keep table | drop table | scales of
    the kept clips]); _plan_maps() turns the buffer into row maps."""
    host = getattr(scale_vec, '_vtx_host', None) if scale_vec is not None else None
    # (table row maps are for groups of >= 256 rows: the GEMM tile maps assume at most one group boundary per 256-row tile and
    # read one spare table entry; a short-sequence model keeps the compute-and-multiply-by-zero path)
    if not _compact or host is None or host.numel() != n_units or rows_per < 256:
        return None
    hv = host.numpy()
    kept = [i for i in range(n_units) if hv[i] != 0.0]
    drop = [i for i in range(n_units) if hv[i] == 0.0]
    if not drop:
        return None
    def table(idx):                                  # one offset per group + the spare entry the tile maps read
        t = [(c - j) * rows_per for j, c in enumerate(idx)]
        return t + [t[-1] if t else 0]
    tk, td = table(kept), table(drop)
    import numpy as np
    words = np.concatenate([np.asarray(tk, np.int32).view(np.float32), np.asarray(td, np.int32).view(np.float32),
                            np.asarray([hv[i] for i in kept], np.float32)])
    buf = ops.upload_f32(torch.from_numpy(words), device)
    step = lambda t: max([b - a for a, b in zip(t, t[1:])] + [0])      # noqa: E731
    return len(kept), len(drop), step(tk), step(td), buf


def _plan_maps(plan, rows_per):
    """(keep map, drop map, per-kept-clip scales) over the plan's device buffer.  Built from the buffer at hand, never
    kept across forward / backward: under activation recomputation (torch.utils.checkpoint) backward receives the
    RE-COMPUTED buffer, and a row map stored by the first forward would point into freed memory."""
    nk, nd, step_k, step_d, buf = plan
    ik = buf[:nk + 1].view(torch.int32)
    idr = buf[nk + 1:nk + nd + 2].view(torch.int32)
    sv_k = buf[nk + nd + 2:]
    kmap = ops.tabmap(rows_per, ik, step_k) if nk else None
    return kmap, ops.tabmap(rows_per, idr, step_d), sv_k


class FFNFn(torch.autograd.Function):
    """FFNWithPreNorm.forward with num_layers == 2 (reference transformer.py:516-523)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, scale_vec, eps=1e-5, xs=None, exact=False):
        x = _chk(x)                                  # exact: the contribution d of the previous sub-block (see TimeAttnFn)
        D = x.shape[-1]
        M = x.numel() // D
        rows_per = M // x.shape[0]
        Hd = w1.shape[0]
        dtp = x.dtype
        plan = _compaction_plan(scale_vec, x.shape[0], rows_per, x.device)
        if plan is not None:
            return FFNFn._forward_compact(ctx, x, ln_w, ln_b, w1, b1, w2, b2, eps, plan, rows_per, xs, exact)
        xn = _empty((M, D), x)
        mean = _empty((M,), x, torch.float32)
        rstd = _empty((M,), x, torch.float32)
        x32 = None
        if exact:
            x32 = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            ops.layernorm_acc_fwd(xs, x, M, D, D, IDENT, x32, D, IDENT, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
        else:
            ops.layernorm_fwd(x, M, D, D, IDENT, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
        w1c, w1T = weights(w1, dtp, any(ctx.needs_input_grad))
        # the second output is gelu'(pre-activation), not the pre-activation: it is all the backward needs of it
        # (transformer.py:503 nn.GELU), computed from the same erf / exp evaluation, and the backward's epilogue
        # becomes one multiply instead of an erf + exp per element
        h = _empty((M, Hd), x)
        g = _empty((M, Hd), x)
        ops.gemm_nt(xn, w1c, g, M, Hd, D, bias=b1, act=2, C2=h)
        w2c, w2T = weights(w2, dtp, any(ctx.needs_input_grad))
        out = torch.empty_like(x)
        ops.gemm_nt(g, w2c, out, M, D, Hd, bias=b2, row_scale=scale_vec, rs=(rows_per, 1, 1, 0), R=None if exact else x)
        ctx.save_for_backward(x32 if exact else x, ln_w, mean, rstd, xn, h, g,
                              scale_vec if scale_vec is not None else x.new_empty(0),
                              *[t for t in (w1T, w2T) if t is not None])
        ctx.cfg = (rows_per, scale_vec is not None, w1.shape[0])
        ctx.params = (ln_w, ln_b, w1, b1, w2, b2)
        if exact:
            ctx.mark_non_differentiable(x32)
            ctx.set_materialize_grads(False)         # no zero-filled 'gradient' of the float32 stream (0.46 GB per sub-block at 96 clips)
            return out, x32
        return out

    @staticmethod
    def _forward_compact(ctx, x, ln_w, ln_b, w1, b1, w2, b2, eps, plan, rows_per, xs=None, exact=False):
        """The kept clips only (see _compaction_plan); the dropped clips leave the block as they came in."""
        nk, nd, step_k, step_d, buf = plan
        kmap, dmap, sv_k = _plan_maps(plan, rows_per)
        D = x.shape[-1]
        Hd = w1.shape[0]
        dtp = x.dtype
        Mk = nk * rows_per
        out = torch.empty_like(x)
        need_t = any(ctx.needs_input_grad)
        w1c, w1T = weights(w1, dtp, need_t)
        w2c, w2T = weights(w2, dtp, need_t)
        x32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if exact else None
        if nk > 0:
            xn = _empty((Mk, D), x)
            mean = _empty((Mk,), x, torch.float32)
            rstd = _empty((Mk,), x, torch.float32)
            if exact:
                ops.layernorm_acc_fwd(xs, x, Mk, D, D, kmap, x32, D, kmap, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
            else:
                ops.layernorm_fwd(x, Mk, D, D, kmap, ln_w, ln_b, eps, xn, D, IDENT, mean, rstd)
            h = _empty((Mk, Hd), x)
            g = _empty((Mk, Hd), x)
            ops.gemm_nt(xn, w1c, g, Mk, Hd, D, bias=b1, act=2, C2=h)
            ops.gemm_nt(g, w2c, out, Mk, D, Hd, cmap=kmap, bias=b2, row_scale=sv_k, rs=(rows_per, 1, 1, 0),
                        R=None if exact else x, rmap=kmap)
        else:
            xn = mean = rstd = h = g = x.new_empty(0)
        if exact:                                    # the dropped clips: the stream moves on, the block contributes nothing
            ops.layernorm_acc_fwd(xs, x, nd * rows_per, D, D, dmap, x32, D, dmap)
            _zero_rows(x, out, nd * rows_per, D, dmap)
        else:
            ops.row_scale_copy(x, out, nd * rows_per, D, smap=dmap, dmap=dmap)
        ctx.save_for_backward(x32 if exact else x, ln_w, mean, rstd, xn, h, g, buf, *[t for t in (w1T, w2T) if t is not None])
        ctx.cfg = (rows_per, True, Hd)
        ctx.plan = (nk, nd, step_k, step_d)
        ctx.params = (ln_w, ln_b, w1, b1, w2, b2)
        if exact:
            ctx.mark_non_differentiable(x32)
            ctx.set_materialize_grads(False)         # no zero-filled 'gradient' of the float32 stream (0.46 GB per sub-block at 96 clips)
            return out, x32
        return out

    @staticmethod
    def _backward_compact(ctx, dout):
        x, ln_w, mean, rstd, xn, h, g, buf, w1T, w2T = ctx.saved_tensors
        p_ln_w, p_ln_b, p_w1, p_b1, p_w2, p_b2 = ctx.params
        rows_per, _, Hd = ctx.cfg
        nk, nd = ctx.plan[:2]
        kmap, dmap, sv_k = _plan_maps(ctx.plan + (buf,), rows_per)
        dout = _chk(dout)
        D = x.shape[-1]
        Mk = nk * rows_per
        x_stream, x = x, dout                          # (the saved stream may be float32: buffers take dout's dtype)
        dx = torch.empty_like(x)
        g32 = _grad_stream(dout, x_stream)
        dx32 = torch.empty_like(g32) if g32 is not None else None
        d_ln_w = d_ln_b = d_w1 = d_b1 = d_w2 = d_b2 = None
        if nk > 0:
            dz = _empty((Mk, D), x)
            ops.row_scale_copy(dout, dz, Mk, D, smap=kmap, s=sv_k, rs=(rows_per, 1, 1, 0))
            d_w2, d_b2 = _linear_grads(p_w2, p_b2, dz, g, Mk, D, Hd)
            dh = _empty((Mk, Hd), x)
            ops.gemm_nt(dz, w2T, dh, Mk, Hd, D, dgelu_in=h, dgelu_kind=1)
            d_w1, d_b1 = _linear_grads(p_w1, p_b1, dh, xn, Mk, Hd, D)
            dxn = _empty((Mk, D), x)
            ops.gemm_nt(dh, w1T, dxn, Mk, D, Hd)
            d_ln_w, d_ln_b, direct = _ln_grad_buffers(p_ln_w, p_ln_b, D, x.device)
            _ln_bwd_res(dxn, x_stream, kmap, Mk, D, mean, rstd, ln_w, dout, dx, d_ln_w, d_ln_b, g32, dx32)
            if direct:
                _fire(p_ln_w, p_ln_b)
                d_ln_w = d_ln_b = None
        else:                                        # every clip dropped: the parameters get no gradient from this block
            for p in (p_ln_w, p_ln_b, p_w1, p_b1, p_w2, p_b2):
                if _sink(p) is not None:
                    _fire(p)
            zeros = [None if _sink(p) is not None else torch.zeros_like(p) for p in (p_ln_w, p_ln_b, p_w1, p_b1, p_w2, p_b2)]
            d_ln_w, d_ln_b, d_w1, d_b1, d_w2, d_b2 = zeros
        _copy_rows_res(dout, dx, g32, dx32, nd * rows_per, D, dmap)
        _hand_on(dx, dx32)
        return (dx, d_ln_w, d_ln_b, d_w1, d_b1, d_w2, d_b2, None, None, None, None)

    @staticmethod
    def backward(ctx, dout, _dstream=None):
        if getattr(ctx, 'plan', None) is not None:
            return FFNFn._backward_compact(ctx, dout)
        x, ln_w, mean, rstd, xn, h, g, sv, w1T, w2T = ctx.saved_tensors
        p_ln_w, p_ln_b, p_w1, p_b1, p_w2, p_b2 = ctx.params
        rows_per, has_scale, Hd = ctx.cfg
        dout = _chk(dout)
        D = x.shape[-1]
        M = x.numel() // D
        dtp = dout.dtype
        x_stream, x = x, dout                          # (the saved stream may be float32: buffers take dout's dtype)
        if has_scale:
            dz = _empty((M, D), x)
            ops.row_scale_copy(dout, dz, M, D, s=sv, rs=(rows_per, 1, 1, 0))
        else:
            dz = dout
        d_w2, d_b2 = _linear_grads(p_w2, p_b2, dz, g, M, D, Hd)
        dh = _empty((M, Hd), x)
        ops.gemm_nt(dz, w2T, dh, M, Hd, D, dgelu_in=h, dgelu_kind=1)
        d_w1, d_b1 = _linear_grads(p_w1, p_b1, dh, xn, M, Hd, D)
        dxn = _empty((M, D), x)
        ops.gemm_nt(dh, w1T, dxn, M, D, Hd)
        dx = torch.empty_like(x)
        g32 = _grad_stream(dout, x_stream)
        dx32 = torch.empty_like(g32) if g32 is not None else None
        d_ln_w, d_ln_b, direct = _ln_grad_buffers(p_ln_w, p_ln_b, D, x.device)
        _ln_bwd_res(dxn, x_stream, IDENT, M, D, mean, rstd, ln_w, dout, dx, d_ln_w, d_ln_b, g32, dx32)
        if direct:
            _fire(p_ln_w, p_ln_b)
            d_ln_w = d_ln_b = None
        _hand_on(dx, dx32)
        return (dx, d_ln_w, d_ln_b, d_w1, d_b1, d_w2, d_b2, None, None, None, None)


class TokensFn(torch.autograd.Function):
    """PatchEmbed + prepare_tokens fused (reference transformer.py:138-151 and
    video_transformer.py:193-240 / :455-502, use_cls_token_temporal=False):
    clip [B,T,C,H,W] fp32 -> residual stream.

    layout 'pt': [B, 1+P*T', D], token 1+p*T'+t'  (divided / joint attention)
    layout 'tp': [(B T'), 1+P, D]                 (space_only, ViViT fact_encoder)"""

    @staticmethod
    def forward(ctx, clip, conv_w, conv_b, cls_token, pos_embed, time_embed, dtype, layout, exact=False):
        # exact (the exact residual stream, bf16 kernels): the stream STARTS in float32 as it does in the reference under autocast --
        # the patch projection is a bf16 tensor (bf16(acc + bias): the autocast Conv's output), the cat with the float32 cls token
        # and the `+ pos_embed`, `+ time_embed` adds promote to float32 (video_transformer.py:200-239).  Returns (d0, xs0): d0 the
        # bf16 projection rows (cls rows 0), xs0 the float32 rest (pos + time; cls rows cls_token + pos[0]); xs0 + d0 is the stream.
        ops.need_cuda(clip, conv_w)
        B, T, Cc, H, W = ops.clip_dims(clip)
        D = conv_w.shape[0]
        ps = conv_w.shape[-1]
        ts = conv_w.shape[2] if conv_w.ndim == 5 else 1
        Tq = T // ts
        P = (H // ps) * (W // ps)
        K = Cc * ts * ps * ps
        # the embedding tables are indexed by patch and frame: a clip with another grid or frame count than the
        # model was built for would read out of bounds (the reference raises a broadcast error here)
        if pos_embed.shape[-2] != 1 + P:
            raise ValueError(f'clip has {P} patches per frame but pos_embed holds {pos_embed.shape[-2] - 1}')
        if time_embed is not None and layout == 'pt' and time_embed.shape[-2] != Tq:
            raise ValueError(f'clip has {Tq} (tubelet) frames but time_embed holds {time_embed.shape[-2]}')
        if T % ts:
            raise ValueError(f'{T} frames are not a multiple of the tubelet size {ts}')
        wc, _ = weights(conv_w, dtype, False)
        exact = bool(exact) and dtype == torch.bfloat16
        tdt = torch.float32 if exact else dtype            # the embedding table: float32 and without the conv bias under the exact stream
        if layout == 'pt':
            rows = ops.patch_rows(clip, dtype, ps, ts, frame_major=False)        # [(b p t), K]
            N = P * Tq
            E, cls_row = ops.embed_table(tdt, P, Tq, D, None if exact else conv_b, pos_embed.reshape(-1, D),
                                         None if time_embed is None else time_embed.reshape(-1, D),
                                         cls_token.reshape(-1))
            nseq, per = B, N
        else:
            rows = ops.patch_rows(clip, dtype, ps, ts, frame_major=True)         # [(b t p), K]
            N = P
            E, cls_row = ops.embed_table(tdt, P, 1, D, None if exact else conv_b, pos_embed.reshape(-1, D), None,
                                         cls_token.reshape(-1))
            nseq, per = B * Tq, P
        x = torch.empty(nseq, 1 + N, D, dtype=dtype, device=clip.device)
        ctx.save_for_backward(rows)
        ctx.cfg = (B, Tq, P, D, K, layout, conv_w.shape, pos_embed.shape,
                   None if time_embed is None else time_embed.shape, cls_token.shape)
        if exact:
            ops.gemm_nt(rows, wc, x, nseq * N, D, K, cmap=ops.tokmap(N), bias=conv_b)
            ops.row_scale_copy(torch.zeros(D, dtype=dtype, device=clip.device), x, nseq, D, smap=ops.rowmap(1, -1, 0), dmap=ops.clsmap(per))
            xs0 = torch.empty(nseq, 1 + N, D, dtype=torch.float32, device=clip.device)
            xs0[:, 1:].copy_(E.view(1, N, D).expand(nseq, N, D))
            xs0[:, 0].copy_(cls_row.view(1, D).expand(nseq, D))
            ctx.mark_non_differentiable(xs0)
            ctx.set_materialize_grads(False)
            return x, xs0
        ops.gemm_nt(rows, wc, x, nseq * N, D, K, cmap=ops.tokmap(N), R=E, r_period=N)
        # cls rows: every sequence's row 0 = cls_token + pos_embed[0]
        ops.row_scale_copy(cls_row, x, nseq, D, smap=ops.rowmap(1, -1, 0), dmap=ops.clsmap(per))
        return x

    @staticmethod
    def backward(ctx, dx, _dxs=None):
        (rows,) = ctx.saved_tensors
        B, Tq, P, D, K, layout, w_shape, pos_shape, time_shape, cls_shape = ctx.cfg
        dx = _chk(dx)
        if layout == 'pt':
            N, nseq = P * Tq, B
        else:
            N, nseq = P, B * Tq
        d_w = ops.gemm_tn(dx, rows, nseq * N, D, K, amap=ops.tokmap(N)).reshape(w_shape)
        # dE[n,:] = sum over sequences of dx[s, 1+n, :]
        dE = ops.reduce_rows(dx, N, nseq, D, D, 1, 1 + N, 1)
        d_b = ops.reduce_rows(dE, 1, N, D, D, 0, 1, 0).reshape(D)
        d_cls = ops.reduce_rows(dx, 1, nseq, D, D, 0, 1 + N, 0)                   # [1, D]
        d_pos = torch.empty(1 + P, D, dtype=torch.float32, device=dx.device)
        d_pos[0:1].copy_(d_cls)
        d_time = None
        if layout == 'pt':
            ops.reduce_rows(dE, P, Tq, D, D, 0, 1, Tq, out=d_pos[1:])
            if time_shape is not None:
                d_time = ops.reduce_rows(dE, Tq, P, D, D, 0, Tq, 1).reshape(time_shape)
        else:
            d_pos[1:].copy_(dE)
        return (None, d_w, d_b, d_cls.reshape(cls_shape), d_pos.reshape(pos_shape), d_time, None, None, None)


class PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed.forward alone (reference transformer.py:138-151): [B,T,C,H,W] ->
    [(B T'), P, D] = conv projection + bias, reference row order."""

    @staticmethod
    def forward(ctx, clip, conv_w, conv_b, dtype):
        ops.need_cuda(clip, conv_w)
        B, T, Cc, H, W = ops.clip_dims(clip)
        D = conv_w.shape[0]
        ps = conv_w.shape[-1]
        ts = conv_w.shape[2] if conv_w.ndim == 5 else 1
        P = (H // ps) * (W // ps)
        K = Cc * ts * ps * ps
        rows = ops.patch_rows(clip, dtype, ps, ts, frame_major=True)
        wc, _ = weights(conv_w, dtype, False)
        M = rows.shape[0]
        y = torch.empty(M, D, dtype=dtype, device=clip.device)
        ops.gemm_nt(rows, wc, y, M, D, K, bias=conv_b)
        ctx.save_for_backward(rows)
        ctx.cfg = (M, D, K, conv_w.shape)
        return y.reshape(B * (T // ts), P, D)

    @staticmethod
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        M, D, K, w_shape = ctx.cfg
        dy = _chk(dy).reshape(M, D)
        d_w, d_b = ops.gemm_tn(dy, rows, M, D, K, want_colsum=True)
        d_w = d_w.reshape(w_shape)
        return None, d_w, d_b, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim of x [.., D]; ``cls_only`` normalises only row 0
    of every [1+N, D] sequence and returns [B, D] (final norm + x[:, 0],
    reference video_transformer.py:251-254 / :527-530).  exact (the exact residual stream, see TimeAttnFn): x is the last
    sub-block's contribution, xs the float32 stream it was computed from; the rows are normalised as xs + x."""

    @staticmethod
    def forward(ctx, x, w, b, eps, cls_only, xs=None, exact=False):
        x = _chk(x)
        D = x.shape[-1]
        if cls_only:
            B, N1, _ = x.shape
            rows, xmap = B, ops.clsmap(N1 - 1)
            y = _empty((B, D), x)
        else:
            rows, xmap = x.numel() // D, IDENT
            y = torch.empty_like(x)
        mean = _empty((rows,), x, torch.float32)
        rstd = _empty((rows,), x, torch.float32)
        if exact:
            x32 = torch.empty(rows, D, dtype=torch.float32, device=x.device)      # the normalised rows of the stream, compact
            ops.layernorm_acc_fwd(xs, x, rows, D, D, xmap, x32, D, IDENT, w, b, eps, y, D, IDENT, mean, rstd)
            ctx.save_for_backward(x32, w, mean, rstd)
        else:
            ops.layernorm_fwd(x, rows, D, D, xmap, w, b, eps, y, D, IDENT, mean, rstd)
            ctx.save_for_backward(x, w, mean, rstd)
        ctx.cfg = (rows, cls_only, exact, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        rows, cls_only, exact, shape = ctx.cfg
        dy = _chk(dy)
        D = x.shape[-1]
        d_w = torch.zeros(D, dtype=torch.float32, device=x.device)
        d_b = torch.zeros(D, dtype=torch.float32, device=x.device)
        if exact:
            # x: the compact float32 rows; their gradient goes back to the rows' places in the stream layout
            dxc = torch.empty(rows, D, dtype=dy.dtype, device=dy.device)
            ops.layernorm_bwd(dy, D, IDENT, x, D, IDENT, rows, D, mean, rstd, w, None, dxc, D, d_w, d_b)
            if cls_only:
                dx = torch.zeros(shape, dtype=dy.dtype, device=dy.device)
                ops.row_scale_copy(dxc, dx, rows, D, smap=IDENT, dmap=ops.clsmap(shape[1] - 1))
            else:
                dx = dxc.view(shape)
            return dx, d_w, d_b, None, None, None, None
        xmap = ops.clsmap(x.shape[1] - 1) if cls_only else IDENT
        dx = torch.zeros_like(x) if cls_only else torch.empty_like(x)
        ops.layernorm_bwd(dy, D, IDENT, x, D, xmap, rows, D, mean, rstd, w, None, dx, D, d_w, d_b)
        return dx, d_w, d_b, None, None, None, None


def _pad8(n):
    return (n + 7) // 8 * 8


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b on [M, K] rows (decoder_pred / classification head).  The GEMM kernels move
    16-byte vectors, so an output width that is not a multiple of 8 (174 / 101 / 51-class heads,
    MaskFeat's default feature_dim=10) runs zero-padded to the next multiple and is sliced back
    (padding / slicing of these [rows, classes]-sized tensors is plain tensor plumbing)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _chk(x)
        K = x.shape[-1]
        M = x.numel() // K
        N = w.shape[0]
        N8 = _pad8(N)
        need_t = any(ctx.needs_input_grad)
        if N8 == N:
            wc, wT = weights(w, x.dtype, need_t)
            bias = b
        else:
            w_pad = torch.zeros(N8, K, dtype=torch.float32, device=w.device)
            w_pad[:N].copy_(w.detach().reshape(N, K))
            wc, wT = ops.cast_transpose(w_pad, x.dtype, want_c=True, want_t=need_t)
            bias = None
            if b is not None:
                bias = torch.zeros(N8, dtype=torch.float32, device=w.device)
                bias[:N].copy_(b.detach())
        y = _empty((M, N8), x)
        ops.gemm_nt(x, wc, y, M, N8, K, bias=bias)
        ctx.save_for_backward(x, *([wT] if wT is not None else []))
        ctx.has_bias = b is not None
        ctx.N = N
        if N8 != N:
            y = y[:, :N]
        return y.reshape(tuple(x.shape[:-1]) + (N,))

    @staticmethod
    def backward(ctx, dy):
        x, wT = ctx.saved_tensors
        K = x.shape[-1]
        M = x.numel() // K
        N = ctx.N
        N8 = _pad8(N)
        dy = dy.reshape(M, N)
        if N8 != N:
            dy = torch.nn.functional.pad(dy, (0, N8 - N))
        dy = _chk(dy)
        if ctx.has_bias:
            d_w, d_b = ops.gemm_tn(dy, x, M, N8, K, want_colsum=True)
            d_b = d_b[:N]
        else:
            d_w, d_b = ops.gemm_tn(dy, x, M, N8, K), None
        dx = torch.empty_like(x)
        ops.gemm_nt(dy, wT, dx, M, K, N8)
        return dx, d_w[:N], d_b


class FactGlueFn(torch.autograd.Function):
    """ViViT fact_encoder: spatial-encoder output x [(b t), 1 + P, D] -> temporal-encoder input [b, 1 + T, D]: the cls rows are
    the first b rows of the flattened (b t) axis (the reference's `x[:b, 0]`, video_transformer.py:515, kept literal), a frame's
    token is the mean over its patches (:516-517), plus time_embed (:519-523).  One kernel forward (vtx_fact_glue_fwd), one + the
    time_embed reduction backward -- round 3 did this with cast / reshape / mean / cat / add in ATen."""

    @staticmethod
    def forward(ctx, x, time_embed, b):
        x = _chk(x)
        BT, P1, D = x.shape
        T = BT // b
        ctx.cfg = (b, T, P1 - 1, D, time_embed.shape)
        return ops.fact_glue_fwd(x, time_embed.reshape(1 + T, D), b, T, P1 - 1, D)

    @staticmethod
    def backward(ctx, dh):
        b, T, P, D, e_shape = ctx.cfg
        dh = _chk(dh)
        need_e = ctx.needs_input_grad[1]
        de = torch.empty(1 + T, D, dtype=torch.float32, device=dh.device) if need_e else None
        if _exact_grad and dh.dtype == torch.bfloat16:
            # 'fp32+grad': a frame token's gradient / P goes to ALL P patch rows of the frame -- rounded to bf16 here, the same rounding
            # error would sit on 196 rows at once and survive every weight-gradient sum over them.  The float32 gradient of the
            # temporal stream in, the float32 gradient of the spatial stream out (its bf16 rounding is what the GEMMs read).
            h = getattr(dh, '_vtx_g32', None)
            g32 = h[0] if (h is not None and h[1] == dh._version and h[0].shape == dh.shape) else ops.cast_to_f32(dh)
            dx32 = ops.fact_glue_bwd(g32, b, T, P, D, d_time_embed=de)
            dx = _hand_on(ops.cast_from_f32(dx32, dh.dtype), dx32)
        else:
            dx = ops.fact_glue_bwd(dh, b, T, P, D, d_time_embed=de)
        return dx, (de.reshape(e_shape) if need_e else None), None


class StreamValueFn(torch.autograd.Function):
    """The exact residual stream as ONE bf16 tensor, for consumers outside the blocks (the ViViT fact-encoder glue, the
    space_only frame mean): bf16(xs + d).  One rounding that nothing accumulates on; d(xs + d)/dd = 1."""

    @staticmethod
    def forward(ctx, d, xs):
        d = _chk(d)
        D = d.shape[-1]
        rows = d.numel() // D
        x32 = torch.empty(d.shape, dtype=torch.float32, device=d.device)
        ops.layernorm_acc_fwd(xs, d, rows, D, D, IDENT, x32, D, IDENT)
        return ops.cast_from_f32(x32, d.dtype)

    @staticmethod
    def backward(ctx, dy):
        return dy, None


class CastFn(torch.autograd.Function):
    """dtype boundary of the path: compute dtype <-> float32."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        if x.dtype == dtype:
            return x
        return ops.cast_to_f32(x) if dtype == torch.float32 else ops.cast_from_f32(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        if dy.dtype == ctx.src:
            return dy, None
        dy = _chk(dy)
        return (ops.cast_to_f32(dy) if ctx.src == torch.float32 else ops.cast_from_f32(dy, ctx.src)), None


class AttnCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v on qkv [Bn, L, 3D] (reference transformer.py:167-174).
    Returns (ctx [Bn, L, D], probs fp32 [Bn, H, L, L] or None)."""

    @staticmethod
    def forward(ctx, qkv, heads, want_probs):
        qkv = _chk(qkv)
        Bn, L, D3 = qkv.shape
        D = D3 // 3
        hd = D // heads
        o = _empty((Bn, L, D), qkv)
        lse = _empty((Bn * heads * L,), qkv, torch.float32)
        probs = _empty((Bn, heads, L, L), qkv, torch.float32) if want_probs else None
        ops.attn_fwd(qkv, o, lse, ATTN_CONTIG, Bn, L, heads, hd, hd ** -0.5, probs=probs)
        ctx.save_for_backward(qkv, o, lse)
        ctx.heads = heads
        if want_probs:
            ctx.mark_non_differentiable(probs)
            return o, probs
        return o, None

    @staticmethod
    def backward(ctx, do, _dprobs):
        qkv, o, lse = ctx.saved_tensors
        heads = ctx.heads
        do = _chk(do)
        Bn, L, D3 = qkv.shape
        hd = D3 // 3 // heads
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(qkv, o, lse, do, dqkv, ATTN_CONTIG, Bn, L, heads, hd, hd ** -0.5)
        return dqkv, None, None


class RowScaleFn(torch.autograd.Function):
    """x * s[row // rows_per] (standalone DropPath, reference transformer.py:34-42)."""

    @staticmethod
    def forward(ctx, x, s, rows_per):
        x = _chk(x)
        D = x.shape[-1]
        y = torch.empty_like(x)
        ops.row_scale_copy(x, y, x.numel() // D, D, s=s, rs=(rows_per, 1, 1, 0))
        ctx.save_for_backward(s)
        ctx.rows_per = rows_per
        return y

    @staticmethod
    def backward(ctx, dy):
        (s,) = ctx.saved_tensors
        dy = _chk(dy)
        D = dy.shape[-1]
        dx = torch.empty_like(dy)
        ops.row_scale_copy(dy, dx, dy.numel() // D, D, s=s, rs=(ctx.rows_per, 1, 1, 0))
        return dx, None, None


class SoftmaxXentFn(torch.autograd.Function):
    """mean_b sum_c -t[b,c] log_softmax(logits[b])[c]: ``target`` is an int64 label vector [B]
    (nn.CrossEntropyLoss, reference model_trainer.py:91) or a float [B,C] soft-target matrix (timm's
    SoftTargetCrossEntropy after Mixup, :87-88).  fp32 logits.  Labels outside [0, C) -- nn.CrossEntropyLoss's default
    ignore_index = -100 -- contribute no loss and no gradient and are left out of the mean's denominator (decided on the
    device: no host synchronisation)."""

    @staticmethod
    def forward(ctx, logits, target):
        logits = _chk(logits.float())
        ops.need_cuda(target)
        B, Cn = logits.shape
        soft = target.is_floating_point()
        if soft:
            if tuple(target.shape) != (B, Cn):
                raise ValueError(f'soft targets must be [{B}, {Cn}], got {tuple(target.shape)}')
            target = target.float().contiguous()
        else:
            if tuple(target.shape) != (B,):
                raise ValueError(f'labels must be [{B}], got {tuple(target.shape)}')
            target = target.long().contiguous()
        rows = torch.empty(B, dtype=torch.float32, device=logits.device)
        lse = torch.empty(B, dtype=torch.float32, device=logits.device)
        mean = torch.empty(2, dtype=torch.float32, device=logits.device)       # [mean, counted rows]
        ops.call('vtx_softmax_xent_fwd', ops.ptr(logits), ops.ptr(target) if soft else None, None if soft else ops.ptr(target),
                 B, Cn, ops.ptr(rows), ops.ptr(lse), ops.ptr(mean), ops.stream())
        ctx.save_for_backward(logits, target, lse, mean)
        ctx.soft = soft
        return mean[0]

    @staticmethod
    def backward(ctx, gloss):
        logits, target, lse, mean = ctx.saved_tensors
        B, Cn = logits.shape
        d = torch.empty_like(logits)
        gloss = gloss.float().contiguous()          # stays on the device: no host sync inside backward
        ops.call('vtx_softmax_xent_bwd', ops.ptr(logits), ops.ptr(target) if ctx.soft else None,
                 None if ctx.soft else ops.ptr(target), ops.ptr(lse), B, Cn, 1.0, ops.ptr(gloss), mean.data_ptr() + 4, ops.ptr(d),
                 ops.stream())
        return d, None


# ---------------------------------------------------------------------------------
# MViT backbone operators (csrc/mvit.hip; reference video_transformer.py:621-800 via pytorchvideo)
# ---------------------------------------------------------------------------------
def _pool_desc(x, thw, heads, stride):
    from ._lib import PoolDesc
    B, N1, Cn = x.shape
    T, H, W = thw
    if N1 != 1 + T * H * W:
        raise ValueError(f'{N1 - 1} tokens do not form a {T}x{H}x{W} grid')
    if stride[0] != 1:
        raise NotImplementedError('vtx: pooling with a temporal stride is not implemented (MViT-B pools space only)')
    d = PoolDesc()
    d.dtype = ops.dt(x); d.B, d.T, d.H, d.W = B, T, H, W
    d.heads, d.hd, d.sh, d.sw = heads, Cn // heads, int(stride[1]), int(stride[2])
    return d


def pooled_thw(thw, stride):
    return [thw[0], (thw[1] - 1) // stride[1] + 1, (thw[2] - 1) // stride[2] + 1]


class PoolConvLNFn(torch.autograd.Function):
    """Pooling of q / k / v in MultiScaleAttention: depthwise Conv3d(3x3x3, stride, padding 1) per head on the
    token grid + LayerNorm(head_dim); cls token bypasses the conv.  x [B, 1+T*H*W, heads*hd]."""

    @staticmethod
    def forward(ctx, x, conv_w, ln_w, ln_b, thw, heads, stride, eps):
        import ctypes as C
        x = _chk(x)
        d = _pool_desc(x, thw, heads, stride)
        if tuple(conv_w.shape[2:]) != (3, 3, 3):
            raise NotImplementedError('vtx: pooling kernels other than 3x3x3 are not implemented')
        T, Ho, Wo = pooled_thw(thw, stride)
        B, _, Cn = x.shape
        n_out = 1 + T * Ho * Wo
        pre = _empty((B, n_out, Cn), x)
        y = _empty((B, n_out, Cn), x)
        mean = _empty((B * n_out * heads,), x, torch.float32)
        rstd = _empty((B * n_out * heads,), x, torch.float32)
        w = conv_w.detach().reshape(d.hd, 27).contiguous()
        ops.call('vtx_pool_conv_ln_fwd', C.byref(d), ops.ptr(x), ops.ptr(w), ops.ptr(ln_w), ops.ptr(ln_b), float(eps),
                 ops.ptr(pre), ops.ptr(y), ops.ptr(mean), ops.ptr(rstd), ops.stream())
        ctx.save_for_backward(x, pre, mean, rstd, w, ln_w)
        ctx.cfg = (tuple(thw), heads, tuple(stride), tuple(conv_w.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        x, pre, mean, rstd, w, ln_w = ctx.saved_tensors
        thw, heads, stride, w_shape = ctx.cfg
        dy = _chk(dy)
        d = _pool_desc(x, thw, heads, stride)
        dpre = torch.empty_like(pre)
        dx = torch.empty_like(x)
        dw = torch.empty(d.hd, 27, dtype=torch.float32, device=x.device)
        dg = torch.empty(d.hd, dtype=torch.float32, device=x.device)
        db = torch.empty(d.hd, dtype=torch.float32, device=x.device)
        ws_bytes = _lib_load().vtx_pool_conv_ln_bwd_workspace(C.byref(d))
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        ops.call('vtx_pool_conv_ln_bwd', C.byref(d), ops.ptr(dy), ops.ptr(x), ops.ptr(pre), ops.ptr(mean), ops.ptr(rstd),
                 ops.ptr(w), ops.ptr(ln_w), ops.ptr(dpre), ops.ptr(dx), ops.ptr(dw), ops.ptr(dg), ops.ptr(db), ops.ptr(ws),
                 ws_bytes, ops.stream())
        return dx, dw.reshape(w_shape), dg, db, None, None, None, None


def _lib_load():
    from . import _lib
    return _lib.load()


class MaxPoolSkipFn(torch.autograd.Function):
    """Residual-path MaxPool3d(kernel (1,3,3), stride (1,2,2), padding (0,1,1)) of MultiScaleBlock; cls row kept."""

    @staticmethod
    def forward(ctx, x, thw):
        x = _chk(x)
        B, N1, Cn = x.shape
        T, H, W = thw
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = _empty((B, 1 + T * Ho * Wo, Cn), x)
        arg = torch.empty(B, 1 + T * Ho * Wo, Cn, dtype=torch.uint8, device=x.device)
        ops.call('vtx_maxpool_skip_fwd', ops.dt(x), B, T, H, W, Cn, ops.ptr(x), ops.ptr(y), ops.ptr(arg), ops.stream())
        ctx.save_for_backward(arg)
        ctx.cfg = (B, T, H, W, Cn, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, T, H, W, Cn, dtype = ctx.cfg
        dy = _chk(dy)
        dx = torch.empty(B, 1 + T * H * W, Cn, dtype=dtype, device=dy.device)
        ops.call('vtx_maxpool_skip_bwd', ops.dt(dy), B, T, H, W, Cn, ops.ptr(dy), ops.ptr(arg), ops.ptr(dx), ops.stream())
        return dx, None


class XAttnFn(torch.autograd.Function):
    """softmax(q k^T hd^-0.5) v with separately pooled q [B,Lq,C] and k, v [B,Lk,C] (heads interleaved in C)."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        from ._lib import XAttnDesc
        q, k, v = _chk(q), _chk(k), _chk(v)
        B, Lq, Cn = q.shape
        out = torch.empty_like(q)
        lse = torch.empty(B * heads * Lq, dtype=torch.float32, device=q.device)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.heads = heads
        XAttnFn._call('vtx_xattn_fwd', q, k, v, out, lse, heads)
        return out

    @staticmethod
    def _desc(q, k, v, out, lse, heads):
        from ._lib import XAttnDesc
        d = XAttnDesc()
        B, Lq, Cn = q.shape
        d.dtype = ops.dt(q); d.B, d.Lq, d.Lk, d.heads, d.hd = B, Lq, k.shape[1], heads, Cn // heads
        d.scale = float((Cn // heads) ** -0.5)
        d.q, d.k, d.v, d.out, d.lse = ops.ptr(q), ops.ptr(k), ops.ptr(v), ops.ptr(out), ops.ptr(lse)
        return d

    @staticmethod
    def _call(name, q, k, v, out, lse, heads, *extra):
        import ctypes as C
        d = XAttnFn._desc(q, k, v, out, lse, heads)
        ops.call(name, C.byref(d), *extra, ops.stream())

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dout = _chk(dout)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        import ctypes as C
        ws_bytes = _lib_load().vtx_xattn_bwd_workspace(C.byref(XAttnFn._desc(q, k, v, out, lse, ctx.heads)))
        ws = torch.empty(max(ws_bytes // 4, 4), dtype=torch.float32, device=q.device)
        XAttnFn._call('vtx_xattn_bwd', q, k, v, out, lse, ctx.heads, ops.ptr(dout), ops.ptr(delta), ops.ptr(dq), ops.ptr(dk),
                      ops.ptr(dv), ops.ptr(ws), ws_bytes)
        return dq, dk, dv, None


class PosEncodingFn(torch.autograd.Function):
    """SpatioTemporalClsPositionalEncoding (separable): cls token prepended, pos_embed_spatial repeated over T +
    pos_embed_temporal repeat-interleaved over H*W + pos_embed_class."""

    @staticmethod
    def forward(ctx, x, cls, pos_class, spatial, temporal):
        x = _chk(x)
        B, N, Cn = x.shape
        T, HW = temporal.shape[-2], spatial.shape[-2]
        if N != T * HW:
            raise ValueError(f'{N} tokens but the position tables cover {T} x {HW}')
        out = _empty((B, 1 + N, Cn), x)
        ops.call('vtx_pos_encoding_fwd', ops.dt(x), B, T, HW, Cn, ops.ptr(x), ops.ptr(cls.detach().reshape(-1).contiguous()),
                 ops.ptr(pos_class.detach().reshape(-1).contiguous()), ops.ptr(spatial.detach().reshape(HW, Cn).contiguous()),
                 ops.ptr(temporal.detach().reshape(T, Cn).contiguous()), ops.ptr(out), ops.stream())
        ctx.cfg = (B, T, HW, Cn, cls.shape, pos_class.shape, spatial.shape, temporal.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, T, HW, Cn, cls_s, pc_s, sp_s, tp_s = ctx.cfg
        dy = _chk(dy)
        N1 = 1 + T * HW
        dx = dy[:, 1:].contiguous()
        d_cls = ops.reduce_rows(dy, 1, B, Cn, Cn, 0, N1, 0)                                  # sum_b dy[b, 0]
        dE = ops.reduce_rows(dy, T * HW, B, Cn, Cn, 1, N1, 1)                                # [T*HW, C] fp32: sum_b dy[b, 1 + n]
        d_sp = ops.reduce_rows(dE, HW, T, Cn, Cn, 0, HW, 1)                                  # sum_t dE[t*HW + hw]
        d_tp = ops.reduce_rows(dE, T, HW, Cn, Cn, 0, 1, HW)                                  # sum_hw
        return dx, d_cls.reshape(cls_s), d_cls.clone().reshape(pc_s), d_sp.reshape(sp_s), d_tp.reshape(tp_s)


class ConvStemFn(torch.autograd.Function):
    """Overlapping, padded Conv3d patch embedding of the MViT stem (create_conv_patch_embed, reference
    video_transformer.py:585-618, 834-843): im2col gather + GEMM, tokens [B, T'*H'*W', D] out.  clip [B,T,C,H,W]."""

    @staticmethod
    def forward(ctx, clip, conv_w, conv_b, stride, padding, dtype):
        import ctypes as C
        ops.need_cuda(clip, conv_w)
        clip = clip.float().contiguous()
        B, T, Cc, H, W = clip.shape
        D, _, KT, KH, KW = conv_w.shape
        K = Cc * KT * KH * KW
        Kp = (K + 63) // 64 * 64                       # whole 64-deep K tiles: the LDS-DMA GEMM kernels apply
        To = (T + 2 * padding[0] - KT) // stride[0] + 1
        Ho = (H + 2 * padding[1] - KH) // stride[1] + 1
        Wo = (W + 2 * padding[2] - KW) // stride[2] + 1
        M = B * To * Ho * Wo
        rows = torch.empty(M, Kp, dtype=dtype, device=clip.device)
        i3 = lambda t: (C.c_int * 3)(*[int(v) for v in t])   # noqa: E731
        from . import ops as _o
        ops.call('vtx_im2col3d', _o._DT[dtype], B, T, Cc, H, W, i3((KT, KH, KW)), i3(stride), i3(padding), Kp, ops.ptr(clip),
                 ops.ptr(rows), ops.stream())
        w_pad = torch.zeros(D, Kp, dtype=torch.float32, device=conv_w.device)
        w_pad[:, :K].copy_(conv_w.detach().reshape(D, K))
        wc, _ = ops.cast_transpose(w_pad, dtype, want_c=True, want_t=False)
        y = torch.empty(M, D, dtype=dtype, device=clip.device)
        ops.gemm_nt(rows, wc, y, M, D, Kp, bias=conv_b)
        ctx.save_for_backward(rows)
        ctx.cfg = (M, D, K, Kp, tuple(conv_w.shape))
        return y.reshape(B, To * Ho * Wo, D)

    @staticmethod
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        M, D, K, Kp, w_shape = ctx.cfg
        dy = _chk(dy).reshape(M, D)
        d_w, d_b = ops.gemm_tn(dy, rows, M, D, Kp, want_colsum=True)
        return None, d_w[:, :K].reshape(w_shape), d_b, None, None, None


class LinearActResFn(torch.autograd.Function):
    """y = act(x W^T + b) (+ res): a Linear with the GELU and / or the residual add of its consumer fused into the
    GEMM epilogue (MViT Mlp.fc1 + GELU; attn.proj / Mlp.fc2 + skip connection).  Output width a multiple of 8."""

    @staticmethod
    def forward(ctx, x, w, b, gelu, res):
        x = _chk(x)
        K = x.shape[-1]
        M = x.numel() // K
        N = w.shape[0]
        wc, wT = weights(w, x.dtype, any(ctx.needs_input_grad))
        y = _empty((M, N), x)
        h = _empty((M, N), x) if gelu else None
        if res is not None:
            res = _chk(res)
        ops.gemm_nt(x, wc, y, M, N, K, bias=b, act=1 if gelu else 0, C2=h, R=res)
        ctx.save_for_backward(x, *( [h] if gelu else []), *([wT] if wT is not None else []))
        ctx.cfg = (gelu, b is not None, res is not None, N)
        return y.reshape(tuple(x.shape[:-1]) + (N,))

    @staticmethod
    def backward(ctx, dy):
        gelu, has_bias, has_res, N = ctx.cfg
        saved = ctx.saved_tensors
        x = saved[0]
        h = saved[1] if gelu else None
        wT = saved[-1]
        K = x.shape[-1]
        M = x.numel() // K
        dy = _chk(dy).reshape(M, N)
        if gelu:                                        # d(pre-activation) = dy * gelu'(h)
            dz = _empty((M, N), x)
            ops.gelu_grad_mul(dy, h, dz)
        else:
            dz = dy
        if has_bias:
            d_w, d_b = ops.gemm_tn(dz, x, M, N, K, want_colsum=True)
        else:
            d_w, d_b = ops.gemm_tn(dz, x, M, N, K), None
        dx = torch.empty_like(x)
        ops.gemm_nt(dz, wT, dx, M, K, N)
        return dx, d_w, d_b, None, (dy.reshape(x.shape[:-1] + (N,)) if has_res else None)


class MlpFn(torch.autograd.Function):
    """y = fc2(GELU(fc1(x))) (+ res): the MViT Mlp as one function.  fc1 writes GELU(h) and GELU'(h) from one evaluation
    (vtx_gemm_nt act = 2), the backward's input-gradient GEMM of fc2 multiplies by the stored derivative in its epilogue
    (as FFNFn does for the TimeSformer block): no separate GELU-gradient pass over the [M, hidden] tensor."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res):
        x = _chk(x)
        K = x.shape[-1]
        M = x.numel() // K
        Hd, N = w1.shape[0], w2.shape[0]
        need_t = any(ctx.needs_input_grad)
        w1c, w1T = weights(w1, x.dtype, need_t)
        w2c, w2T = weights(w2, x.dtype, need_t)
        g = _empty((M, Hd), x)
        gp = _empty((M, Hd), x)
        ops.gemm_nt(x, w1c, g, M, Hd, K, bias=b1, act=2, C2=gp)
        y = _empty((M, N), x)
        if res is not None:
            res = _chk(res)
        ops.gemm_nt(g, w2c, y, M, N, Hd, bias=b2, R=res)
        ctx.save_for_backward(x, g, gp, *[t for t in (w1T, w2T) if t is not None])
        ctx.cfg = (b1 is not None, b2 is not None, res is not None, Hd, N)
        return y.reshape(tuple(x.shape[:-1]) + (N,))

    @staticmethod
    def backward(ctx, dy):
        x, g, gp, w1T, w2T = ctx.saved_tensors
        has_b1, has_b2, has_res, Hd, N = ctx.cfg
        K = x.shape[-1]
        M = x.numel() // K
        dy = _chk(dy).reshape(M, N)
        if has_b2:
            d_w2, d_b2 = ops.gemm_tn(dy, g, M, N, Hd, want_colsum=True)
        else:
            d_w2, d_b2 = ops.gemm_tn(dy, g, M, N, Hd), None
        dh = _empty((M, Hd), x)
        ops.gemm_nt(dy, w2T, dh, M, Hd, N, dgelu_in=gp, dgelu_kind=1)
        if has_b1:
            d_w1, d_b1 = ops.gemm_tn(dh, x, M, Hd, K, want_colsum=True)
        else:
            d_w1, d_b1 = ops.gemm_tn(dh, x, M, Hd, K), None
        dx = torch.empty_like(x)
        ops.gemm_nt(dh, w1T, dx, M, K, Hd)
        return dx, d_w1, d_b1, d_w2, d_b2, (dy.reshape(x.shape[:-1] + (N,)) if has_res else None)
