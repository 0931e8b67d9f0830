"""The full-size MaskFeat / MViT-B parity case (BASELINE cfg 4), shared by tests/test_gpu_mvit.py and the golden generator
tests/golden/make_golden_mvit.py (TEST INFRASTRUCTURE ONLY; parity of the backbone is UNPINNED, see mvit_oracle.py).

One 16x224^2 clip through MaskFeat exactly as the reference's trainer constructs it (model_trainer.py:53-54):
conv stem -> mask-token blend -> MViT-B (16 blocks, 25 089 -> 6 273 -> 1 569 tokens) -> decoder -> HOG-target masked MSE
on the centre frames of the masked cubes.  ``reference()`` is the CPU computation: the head follows the reference's own
lines (video_transformer.py:876-922) with torch ops, the backbone is oracle/mvit_oracle.py."""
import torch

from . import mvit_oracle as MO
from . import synth

MASKFEAT_KW = dict(pool_q_stride_size=[[1, 1, 2, 2], [3, 1, 2, 2]], feature_dim=2 * 2 * 2 * 3 * 9)
HEAD_KEYS = ('patch_embed.patch_model.weight', 'patch_embed.patch_model.bias', 'decoder_pred.weight', 'decoder_pred.bias',
             'mask_token')


def make_oracle():
    return MO.MultiscaleVisionTransformers(embed_dim_mul=[[1, 2.0], [3, 2.0], [14, 2.0]], atten_head_mul=[[1, 2.0], [3, 2.0], [14, 2.0]],
                                           pool_q_stride_size=[[1, 1, 2, 2], [3, 1, 2, 2]], pool_kv_stride_adaptive=[1, 8, 8],
                                           pool_kvq_kernel=[3, 3, 3])


def backbone_state(oracle, seed):
    sd = synth.synth_state_dict(synth.shapes_of(oracle), seed)
    for k in list(sd):                              # depthwise pooling kernels: fan-in 27, keep them O(1)
        if 'pool_' in k:
            sd[k] = sd[k] * 3.0
        if 'pos_embed' in k:
            sd[k] = synth.synth_tensor(k, tuple(sd[k].shape), seed) * 0.02
    return sd


def head_state(seed):
    """Stem / decoder / mask token: synthetic like the backbone (not the constructor's RNG-dependent init)."""
    shapes = {'patch_embed.patch_model.weight': (96, 3, 3, 7, 7), 'patch_embed.patch_model.bias': (96,),
              'decoder_pred.weight': (216, 768), 'decoder_pred.bias': (216,), 'mask_token': (1, 1, 96)}
    return {k: synth.synth_tensor(k, s, seed) for k, s in shapes.items()}


def inputs(B=1):
    x = synth.synth_clip(B, 16, seed=8)
    g = torch.Generator().manual_seed(99)
    target = torch.rand(B, 16, 14, 14, 108, generator=g, dtype=torch.float64)
    mask = torch.zeros(B, 8, 14, 14, dtype=torch.int32)
    mask[0, 2:4, 3:9, 2:10] = 1
    mask[0, 6, 5:12, 5:12] = 1
    markers = [[[2, 2], [6, 1]]] + [[] for _ in range(B - 1)]
    return x, target, mask, markers


def reference(oracle, head, x, target, mask, markers, autocast=False):
    """-> (pred [B,16,14,14,108], loss, {name: grad}) with names as in MaskFeat's state_dict (backbone under 'mvit.').
    float64 throughout; autocast=True: float32 parameters under torch.autocast('cpu', bfloat16) -- the AMP class the
    reference trains in (model_pretrain.py:203) applied to the same graph, the yardstick of the bf16 bars."""
    dt = torch.float32 if autocast else torch.float64
    oracle = oracle.to(dt)
    oracle.zero_grad()
    hp = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in head.items()}
    B = x.shape[0]
    ctx = torch.autocast('cpu', dtype=torch.bfloat16) if autocast else torch.autocast('cpu', enabled=False)
    with ctx:
        tok = torch.nn.functional.conv3d(x.to(dt).transpose(1, 2), hp['patch_embed.patch_model.weight'], hp['patch_embed.patch_model.bias'],
                                         stride=(2, 4, 4), padding=(1, 3, 3)).flatten(2).transpose(1, 2)
        wmask = mask.repeat_interleave(4, 2).repeat_interleave(4, 3).flatten(1).unsqueeze(-1).to(tok.dtype)
        tok = tok * (1 - wmask) + hp['mask_token'].to(tok.dtype) * wmask
        feat = oracle(tok)
        p = torch.nn.functional.linear(feat, hp['decoder_pred.weight'], hp['decoder_pred.bias'])[:, 1:]
        p = p.reshape(B, 8, 14, 14, 2, 108).permute(0, 1, 4, 2, 3, 5).reshape(B, 16, 14, 14, 108)
    mk = mask.repeat_interleave(2, 1).clone()
    for b in range(B):
        keep = torch.zeros(16, dtype=torch.bool)
        for s, span in markers[b]:
            keep[s * 2 + span * 2 // 2] = True
        mk[b, ~keep] = 0
    loss = (((p.double() - target) ** 2).mean(-1) * mk).sum() / (mk.sum() + 1e-5)
    loss.backward()
    grads = {k: v.grad.detach().double() for k, v in hp.items()}
    grads.update({'mvit.' + k: v.grad.detach().double() for k, v in oracle.named_parameters()})
    return p.detach().double(), loss.detach().double(), grads
