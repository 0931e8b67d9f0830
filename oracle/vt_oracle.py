"""fp32 CPU oracle of the video-transformer hot path (TEST INFRASTRUCTURE ONLY).

A functional restatement, over a plain ``state_dict``, of what the reference's
``nn.Module`` graph computes.  Every function cites the reference lines it
follows (paths relative to /root/reference).  It is written with explicit index
arithmetic instead of the reference's einops strings so that the token-order
contract (token index = 1 + p*T + t) is spelled out once, here.

Pinned against the running reference by tests/test_oracle_pin.py and against
the committed golden vectors (tests/golden/) by tests/test_oracle_golden.py.
Autograd through these functions gives the oracle gradients.
"""
import math

import numpy as np
import torch

SQRT1_2 = 0.7071067811865476


# --------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------
def layer_norm(x, weight, bias, eps):
    """nn.LayerNorm over the last dim (transformer.py:215,321,418,495 eps=1e-5;
    video_transformer.py:119,401 eps=1e-6). Biased variance, fp32 statistics."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * weight + bias


def gelu_erf(x):
    """nn.GELU() default = exact erf form (transformer.py:483,502)."""
    return 0.5 * x * (1.0 + torch.erf(x * SQRT1_2))


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def drop_path(x, p, training):
    """transformer.py:34-42.  Draws ``torch.rand((rows,1,..,1))`` from the CPU
    default generator -- and draws nothing at all when p == 0 or in eval."""
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    u = torch.rand((x.shape[0],) + (1,) * (x.ndim - 1))
    mask = torch.floor(keep + u.to(x.dtype))
    return x / keep * mask


def attention(x, sd, pre, heads):
    """transformer.py:165-177.  qkv output features are ordered [3][head][hd]
    (:167); scale = hd**-0.5 (:158); returns (proj(ctx), softmax probs)."""
    bn, n, d = x.shape
    hd = d // heads
    qkv = linear(x, sd[pre + 'qkv.weight'], sd[pre + 'qkv.bias'])
    qkv = qkv.reshape(bn, n, 3, heads, hd)
    q = qkv[:, :, 0].permute(0, 2, 1, 3)          # [bn, h, n, hd]
    k = qkv[:, :, 1].permute(0, 2, 1, 3)
    v = qkv[:, :, 2].permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
    p = torch.softmax(s, dim=-1)
    ctx = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(bn, n, d)
    return linear(ctx, sd[pre + 'proj.weight'], sd[pre + 'proj.bias']), p


# --------------------------------------------------------------------------
# sub-blocks  (x is the residual stream [B, 1 + P*T, D]; token = 1 + p*T + t)
# --------------------------------------------------------------------------
def time_attn_block(x, sd, pre, T, heads, dp, training, return_attention=False):
    """DividedTemporalAttentionWithPreNorm.forward, use_cls_token=False path
    (transformer.py:234-282: cls split :238-244, '(b p) t d' :250, LN :257,
    attention :261, DropPath before temporal_fc :265-267, residual :279-281)."""
    b, n1, d = x.shape
    cls, tok = x[:, :1], x[:, 1:]
    n = n1 - 1
    seq = tok.reshape(b * (n // T), T, d)        # rows (b,p), T contiguous tokens
    h = layer_norm(seq, sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-5)
    a, probs = attention(h, sd, pre + 'attn.', heads)
    if return_attention:
        return probs
    a = drop_path(a, dp, training)
    a = linear(a, sd[pre + 'temporal_fc.weight'], sd[pre + 'temporal_fc.bias'])
    return torch.cat([cls, tok + a.reshape(b, n, d)], dim=1)


def space_attn_block(x, sd, pre, T, heads, dp, training, return_attention=False):
    """DividedSpatialAttentionWithPreNorm.forward, use_cls_token=True path
    (transformer.py:336-382: residual incl. cls :342, '(b t) p d' :352, cls
    replicated per frame :354-356, LN :359, attention :363, DropPath :367,
    cls mean over frames :371-373, scatter back + residual :375-377)."""
    b, n1, d = x.shape
    n = n1 - 1
    P = n // T
    cls, tok = x[:, :1], x[:, 1:]
    frames = tok.reshape(b, P, T, d).permute(0, 2, 1, 3).reshape(b * T, P, d)
    cls_rep = cls.expand(b, T, d).reshape(b * T, 1, d)
    seq = torch.cat([cls_rep, frames], dim=1)    # [(b t), 1+P, d]
    h = layer_norm(seq, sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-5)
    a, probs = attention(h, sd, pre + 'attn.', heads)
    if return_attention:
        return probs
    a = drop_path(a, dp, training)
    cls_out = a[:, 0].reshape(b, T, d).mean(dim=1, keepdim=True)
    tok_out = a[:, 1:].reshape(b, T, P, d).permute(0, 2, 1, 3).reshape(b, n, d)
    return x + torch.cat([cls_out, tok_out], dim=1)


def self_attn_block(x, sd, pre, heads, dp, training, return_attention=False):
    """MultiheadAttentionWithPreNorm.forward (transformer.py:428-456)."""
    h = layer_norm(x, sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-5)
    a, probs = attention(h, sd, pre + 'attn.', heads)
    if return_attention:
        return probs
    return x + drop_path(a, dp, training)


def ffn_block(x, sd, pre, dp, training):
    """FFNWithPreNorm.forward (transformer.py:516-523); two linears, erf-GELU."""
    h = layer_norm(x, sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-5)
    h = gelu_erf(linear(h, sd[pre + 'layers.0.0.weight'], sd[pre + 'layers.0.0.bias']))
    h = linear(h, sd[pre + 'layers.1.weight'], sd[pre + 'layers.1.bias'])
    return x + drop_path(h, dp, training)


def container(x, sd, pre, n_layers, ops, T, heads, training, return_attention=False,
              drop_path_rate=0.1):
    """TransformerContainer / BasicTransformerBlock (transformer.py:526-636).
    dpr = linspace(0, 0.1, L) (:543); with return_attention the last block's
    last attention op returns its probabilities and nothing after runs
    (:560-561, :628-630)."""
    dpr = np.linspace(0, drop_path_rate, n_layers)
    n_attn = sum(1 for o in ops if o != 'ffn')
    for i in range(n_layers):
        lp = f'{pre}layers.{i}.'
        want = return_attention and i >= n_layers - 1
        ai = 0
        for op in ops:
            if op == 'ffn':
                x = ffn_block(x, sd, lp + 'ffns.0.', float(dpr[i]), training)
                continue
            ra = want and ai >= n_attn - 1
            ap = f'{lp}attentions.{ai}.'
            if op == 'time_attn':
                x = time_attn_block(x, sd, ap, T, heads, float(dpr[i]), training, ra)
            elif op == 'space_attn':
                x = space_attn_block(x, sd, ap, T, heads, float(dpr[i]), training, ra)
            elif op == 'self_attn':
                x = self_attn_block(x, sd, ap, heads, float(dpr[i]), training, ra)
            else:
                raise TypeError(op)
            if ra:
                return x
            ai += 1
    return x


# --------------------------------------------------------------------------
# patch / tubelet embedding
# --------------------------------------------------------------------------
def patch_rows_2d(x, ps):
    """[B,T,C,H,W] -> [(B T), P, C*ps*ps]; K index = c*ps*ps + kh*ps + kw, patch
    p = ph*(W/ps) + pw.  Conv2d(k=s=ps) == this gather followed by a GEMM
    (transformer.py:116-120,145-147)."""
    b, t, c, hh, ww = x.shape
    gh, gw = hh // ps, ww // ps
    r = x.reshape(b * t, c, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5)
    return r.reshape(b * t, gh * gw, c * ps * ps)


def patch_rows_3d(x, ps, ts):
    """[B,T,C,H,W] -> [(B T/ts), P, C*ts*ps*ps]; K = c*ts*ps*ps + kt*ps*ps + kh*ps
    + kw (Conv3d weight [D,C,ts,ps,ps], transformer.py:121-126,140-143)."""
    b, t, c, hh, ww = x.shape
    gt, gh, gw = t // ts, hh // ps, ww // ps
    r = x.reshape(b, gt, ts, c, gh, ps, gw, ps).permute(0, 1, 4, 6, 3, 2, 5, 7)
    return r.reshape(b * gt, gh * gw, c * ts * ps * ps)


def patch_embed(x, sd, pre='patch_embed.projection.'):
    w = sd[pre + 'weight']
    if w.ndim == 4:
        rows = patch_rows_2d(x, w.shape[-1])
    else:
        rows = patch_rows_3d(x, w.shape[-1], w.shape[2])
    return linear(rows, w.reshape(w.shape[0], -1), sd[pre + 'bias'])


# --------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------
def interpolated_pos_embed(pos_embed, npatch, w, h, patch):
    """TimeSformer.interpolate_pos_encoding (video_transformer.py:171-191): the table as it is when the clip has the patch grid
    the model was built for and is square, else a bicubic resize of the patch part.  Kept quirks: both grid sides are divided by
    patch_size[0]; the WIDTH ratio scales the first grid axis although the tokens are (h w)-ordered; 0.1 is added to each side
    before the ratio (the DINO float-rounding workaround)."""
    n = pos_embed.shape[1] - 1
    if npatch == n and w == h:
        return pos_embed
    d = pos_embed.shape[-1]
    side = int(math.sqrt(n))
    w0, h0 = w // patch + 0.1, h // patch + 0.1
    grid = pos_embed[:, 1:].reshape(1, side, side, d).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, scale_factor=(w0 / math.sqrt(n), h0 / math.sqrt(n)), mode='bicubic')
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    return torch.cat([pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, d)], dim=1)


def _tokens_with_time(tok, sd, b):
    """Shared tail of prepare_tokens for every non-factorised attention type
    (video_transformer.py:199-238 / :461-500, use_cls_token_temporal=False):
    tok is [(b t), P, D] straight out of the patch embed."""
    bt, P, d = tok.shape
    T = bt // b
    cls = sd['cls_token'].expand(bt, 1, d)
    x = torch.cat([cls, tok], dim=1) + sd['pos_embed']       # [(b t), 1+P, d]
    cls_b = x[:b, 0:1]                                        # x[:b,0,:]  (:216)
    body = x[:, 1:].reshape(b, T, P, d).permute(0, 2, 1, 3)   # [b, P, T, d]
    body = body + sd['time_embed'].reshape(1, 1, T, d)
    return torch.cat([cls_b, body.reshape(b, P * T, d)], dim=1)


def timesformer_forward(sd, x, num_frames, heads=12, layers=12,
                        attention_type='divided_space_time', training=False,
                        return_attention=False, return_cls_token=True):
    """TimeSformer.forward / get_last_selfattention
    (video_transformer.py:193-261)."""
    b = x.shape[0]
    tok = patch_embed(x, sd)
    patch = sd['patch_embed.projection.weight'].shape[-1]
    pe = interpolated_pos_embed(sd['pos_embed'], tok.shape[1], x.shape[-1], x.shape[-2], patch)      # (:209)
    if pe is not sd['pos_embed']:
        sd = dict(sd, pos_embed=pe)
    if attention_type == 'space_only':
        bt, P, d = tok.shape
        h = torch.cat([sd['cls_token'].expand(bt, 1, d), tok], dim=1) + sd['pos_embed']
        ops = ['self_attn', 'ffn']
    else:
        h = _tokens_with_time(tok, sd, b)
        ops = (['time_attn', 'space_attn', 'ffn']
               if attention_type == 'divided_space_time' else ['self_attn', 'ffn'])
    h = container(h, sd, 'transformer_layers.', layers, ops, num_frames, heads,
                  training, return_attention)
    if return_attention:
        return h
    if attention_type == 'space_only':
        h = h.reshape(b, -1, h.shape[1], h.shape[2]).mean(dim=1)
    h = layer_norm(h, sd['norm.weight'], sd['norm.bias'], 1e-6)
    return h[:, 0] if return_cls_token else h[:, 1:].mean(1)


def vivit_forward(sd, x, num_frames, tube_size=2, heads=12, layers=12,
                  attention_type='fact_encoder', training=False,
                  return_cls_token=True):
    """ViViT.forward (video_transformer.py:455-532).  fact_encoder keeps the
    reference's ``x[:b, 0, :]`` read of the flattened (b t) axis (:515)."""
    b = x.shape[0]
    T = num_frames // tube_size
    tok = patch_embed(x, sd)
    if attention_type != 'fact_encoder':
        h = _tokens_with_time(tok, sd, b)
        ops = (['time_attn', 'space_attn', 'ffn']
               if attention_type == 'divided_space_time' else ['self_attn', 'ffn'])
        h = container(h, sd, 'transformer_layers.', layers, ops, T, heads, training)
    else:
        bt, P, d = tok.shape
        h = torch.cat([sd['cls_token'].expand(bt, 1, d), tok], dim=1) + sd['pos_embed']
        h = container(h, sd, 'transformer_layers.0.', layers, ['self_attn', 'ffn'],
                      T, heads, training)
        cls_b = h[:b, 0:1]                                     # the :515 quirk
        frames = h[:, 1:].reshape(b, T, P, d).mean(dim=2)      # [b, T, d]
        h = torch.cat([cls_b, frames], dim=1) + sd['time_embed']
        h = container(h, sd, 'transformer_layers.1.', 4, ['self_attn', 'ffn'],
                      T, heads, training)
    h = layer_norm(h, sd['norm.weight'], sd['norm.bias'], 1e-6)
    return h[:, 0] if return_cls_token else h[:, 1:].mean(1)


# --------------------------------------------------------------------------
# MaskFeat head (video_transformer.py:876-922), backbone excluded
# --------------------------------------------------------------------------
def maskfeat_blend(tokens, mask, mask_token, downsample_rate):
    """forward_features :914-919.  tokens [B, T'*H'*W', C] from the conv patch
    embed (H' = 14*downsample_rate), mask [B,T',14,14] -> nearest upsample."""
    dense = mask.repeat_interleave(downsample_rate, 2).repeat_interleave(downsample_rate, 3)
    w = dense.flatten(1).unsqueeze(-1).to(mask_token.dtype)
    return tokens * (1 - w) + mask_token.expand(tokens.shape[0], tokens.shape[1], -1) * w


def center_frame_mask(mask, cube_marker, num_frames, tstride):
    """:889-896.  mask [B,T',h,w] -> [B,T,h,w] with only the centre frame of
    every cube kept; centre = start*ts + span*ts//2."""
    m = mask.repeat_interleave(tstride, 1).clone()
    keep = torch.zeros(mask.shape[0], num_frames, dtype=torch.bool)
    for i, markers in enumerate(cube_marker):
        for start, span in markers:
            keep[i, start * tstride + span * tstride // 2] = True
    m[~keep] = 0
    return m, keep


def maskfeat_head(feat, w, bias, target, mask, cube_marker, num_frames=16,
                  tstride=2, grid=14):
    """MaskFeat.forward after the backbone (:878-901).  feat [B, 1+T'*g*g, Din];
    pred features split (dt dc) -> frames t*ts+dt.  Loss is float64 because the
    HOG target is (dataset.py:190)."""
    pred = linear(feat, w, bias)[:, 1:]
    b = pred.shape[0]
    tq = num_frames // tstride
    pred = pred.reshape(b, tq, grid, grid, tstride, -1).permute(0, 1, 4, 2, 3, 5)
    pred = pred.reshape(b, num_frames, grid, grid, -1)
    m, keep = center_frame_mask(mask, cube_marker, num_frames, tstride)
    err = ((pred - target) ** 2).mean(dim=-1)
    loss = (err * m).sum() / (m.sum() + 1e-5)
    return pred, loss
