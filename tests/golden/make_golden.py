"""Generate the committed golden vectors from the UNMODIFIED reference.

Runs only in the dev container (needs /root/reference and, for HOG, the real
scikit-image 0.18.3 under /opt/conda/bin/python3.9):

    python tests/golden/make_golden.py

Everything is seeded through oracle/synth.py, so the GPU box (which has neither
the reference nor skimage) can rebuild the exact inputs / weights and compare its
HIP results with the stored reference outputs.
"""
import json
import os
import random
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_loader, synth  # noqa: E402

SMALL = dict(img_size=64, patch_size=16, embed_dims=128, num_heads=2, num_transformer_layers=2)


def loss_weights(d, seed=0):
    return synth.synth_tensor('loss_w', (d,), seed) * 10.0


def grads_summary(model, full_limit=17000):
    out = {}
    for k, p in model.named_parameters():
        g = p.grad
        if g is None:
            continue
        g = g.detach().double()
        if g.numel() <= full_limit:
            out['g:' + k] = g.float().numpy()
        else:
            out['gh:' + k] = g.flatten()[:256].float().numpy()        # head of the tensor
            out['gn:' + k] = np.array([g.norm().item(), g.sum().item()])
    return out


def run_model(model, x, sd, train, seed=None, want_grads=True, heads=None):
    model.load_state_dict(sd, strict=True)
    model.train(train)
    if seed is not None:
        torch.manual_seed(seed)
    res = {}
    if want_grads:
        y = model(x)
        w = loss_weights(y.shape[-1])
        (y * w).sum().backward()
        res.update(grads_summary(model))
    else:
        with torch.no_grad():
            y = model(x)
    res['out'] = y.detach().numpy()
    return res


def main():
    R = ref_loader.load()
    VT = R.video_transformer
    torch.set_num_threads(8)
    # ---- state_dict key contract ---------------------------------------------------
    keys = {}
    for name, ctor in [('timesformer_b_t8', lambda: VT.TimeSformer(num_frames=8)),
                       ('vivit_b_t16', lambda: VT.ViViT(num_frames=16))]:
        keys[name] = {k: list(v.shape) for k, v in ctor().state_dict().items()}
    json.dump(keys, open(os.path.join(HERE, 'state_dict_keys.json'), 'w'), indent=0, sort_keys=True)

    # ---- cfg 1: TimeSformer-B, T=2, B=2, eval forward (BASELINE.json configs[0]) -----
    m = VT.TimeSformer(num_frames=2)
    sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
    r = run_model(m, synth.synth_clip(2, 2, seed=0), sd, train=False, want_grads=False)
    np.savez_compressed(os.path.join(HERE, 'tsf_b_cfg1.npz'), **r)
    print('cfg1', r['out'].shape, float(np.abs(r['out']).max()))

    # ---- cfg 2 shape: TimeSformer-B, T=8, B=1: eval forward + attention map ------------
    m = VT.TimeSformer(num_frames=8)
    sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
    # (the train-mode fwd+bwd golden of this configuration is tsf_b_t8_autocast.npz, make_golden_r2.py)
    r = run_model(m, synth.synth_clip(1, 8, seed=1), sd, train=False, want_grads=False)
    m.zero_grad()
    with torch.no_grad():
        att = m.get_last_selfattention(synth.synth_clip(1, 8, seed=1))
    r['attn_shape'] = np.array(att.shape)
    r['attn_head'] = att[:2, :, :8, :8].numpy()
    r['attn_rowsum'] = att.sum(-1)[:, :, :4].numpy()
    np.savez_compressed(os.path.join(HERE, 'tsf_b_t8_eval.npz'), **r)

    # ---- small TimeSformer, every attention type, all gradients ---------------------
    for at in ['divided_space_time', 'space_only', 'joint_space_time']:
        m = VT.TimeSformer(num_frames=4, attention_type=at, **SMALL)
        sd = synth.synth_state_dict(synth.shapes_of(m), seed=3)
        x = synth.synth_clip(3, 4, 3, 64, 64, seed=2)
        r = run_model(m, x, sd, train=True, seed=11)
        m.zero_grad()
        r_eval = run_model(m, x, sd, train=False, want_grads=False)
        r['out_eval'] = r_eval['out']
        m.eval()
        with torch.no_grad():
            r['attn'] = m.get_last_selfattention(x).numpy()
        np.savez_compressed(os.path.join(HERE, f'tsf_small_{at}.npz'), **r)
        print('tsf_small', at, r['attn'].shape)

    # ---- ViViT: small (all grads, all types) and ViViT-B fact_encoder forward ---------
    for at in ['fact_encoder', 'joint_space_time', 'divided_space_time']:
        m = VT.ViViT(num_frames=8, attention_type=at, **SMALL)
        sd = synth.synth_state_dict(synth.shapes_of(m), seed=4)
        x = synth.synth_clip(3, 8, 3, 64, 64, seed=5)
        r = run_model(m, x, sd, train=True, seed=13)
        np.savez_compressed(os.path.join(HERE, f'vivit_small_{at}.npz'), **r)
    m = VT.ViViT(num_frames=16)
    sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
    r = run_model(m, synth.synth_clip(2, 16, seed=3), sd, train=False, want_grads=False)
    np.savez_compressed(os.path.join(HERE, 'vivit_b_t16_eval.npz'), **r)
    print('vivit_b', r['out'].shape)

    # ---- MaskFeat head through the reference's own forward (shim) ---------
    import torch.nn as nn
    shim = VT.MaskFeat.__new__(VT.MaskFeat)
    nn.Module.__init__(shim)
    shim.num_frames, shim.img_size, shim.stride, shim.downsample_rate = 16, 224, (2, 4, 4), 4
    shim.patch_embed = VT.create_conv_patch_embed(in_channels=3, out_channels=96, conv_kernel_size=(3, 7, 7),
                                                  conv_stride=(2, 4, 4), conv_padding=(1, 3, 3), conv=nn.Conv3d)
    shim.decoder_pred = nn.Linear(768, 216)
    shim.mask_token = nn.Parameter(torch.zeros(1, 1, 96))

    class StandIn(nn.Module):       # [B, 25088, 96] -> [B, 1569, 768]: fixed random projection + 4x4 pooling
        def __init__(self):
            super().__init__()
            self.register_buffer('w', synth.synth_tensor('standin.w', (768, 96), 0))

        def forward(self, t):
            b = t.shape[0]
            v = t.reshape(b, 8, 14, 4, 14, 4, 96).mean(dim=(3, 5)).reshape(b, 1568, 96) @ self.w.t()
            return torch.cat([v.mean(1, keepdim=True), v], dim=1)
    shim.mvit = StandIn()
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in shim.state_dict().items() if 'mvit' not in k}, seed=6)
    shim.load_state_dict(sd, strict=False)
    random.seed(0)
    gen = R.mask_generator.CubeMaskGenerator(input_size=(8, 14, 14), min_num_patches=16)
    B = 2
    masks, markers = [], []
    for _ in range(B):
        mk, cm = gen()
        masks.append(mk)
        markers.append(cm)
    mask = torch.from_numpy(np.stack(masks))
    x = synth.synth_clip(B, 16, seed=8)
    g = torch.Generator().manual_seed(99)
    target = torch.rand(B, 16, 14, 14, 108, generator=g, dtype=torch.float64)
    pred, loss = VT.MaskFeat.forward(shim, x, target, mask.clone(), markers)
    loss.backward()
    tokens = shim.patch_embed(x.transpose(1, 2)).detach()
    blended_ref = (tokens * (1 - mask.repeat_interleave(4, 2).repeat_interleave(4, 3).flatten(1).unsqueeze(-1).float())
                   + shim.mask_token.detach() * mask.repeat_interleave(4, 2).repeat_interleave(4, 3).flatten(1).unsqueeze(-1).float())
    np.savez_compressed(
        os.path.join(HERE, 'maskfeat_head.npz'), mask=mask.numpy(), markers=json.dumps(markers),
        pred_head=pred.detach()[:, :, :2, :2].numpy(), pred_sum=np.array(pred.detach().double().sum().item()),
        loss=np.array(loss.item()), d_decoder_w_head=shim.decoder_pred.weight.grad[:8].numpy(),
        d_decoder_b=shim.decoder_pred.bias.grad.numpy(), d_mask_token=shim.mask_token.grad.numpy(),
        blend_checksum=np.array([blended_ref.double().sum().item(), blended_ref.double().abs().sum().item()]))
    print('maskfeat loss', loss.item(), markers)

    # ---- HOG: real scikit-image 0.18.3 ------------------------------------------------
    code = r'''
import sys, numpy as np
from skimage.feature import hog
from einops import rearrange
def ext(image):
    fs=[hog(image[:,:,c], orientations=9, pixels_per_cell=(8,8), cells_per_block=(1,1), block_norm='L2', feature_vector=False) for c in range(3)]
    return rearrange(np.concatenate(fs,axis=-1),'(ph dh) (pw dw) ch cw c -> ph pw (dh dw ch cw c)',ph=14,pw=14)
frames=[np.random.RandomState(s).randint(0,256,(224,224,3)).astype(np.uint8) for s in (1234, 7)]
yy,xx=np.mgrid[0:224,0:224]
frames.append(np.stack([(yy*255//223),(xx*255//223),((xx*3+yy*5)//8%256)],-1).astype(np.uint8))
np.savez_compressed(sys.argv[1], feats=np.stack([ext(f) for f in frames]), seeds=np.array([1234,7,-1]))
'''
    out = os.path.join(HERE, 'hog_skimage.npz')
    subprocess.run(['/opt/conda/bin/python3.9', '-W', 'ignore', '-c', code, out], check=True)
    f = np.load(out)['feats']
    print('hog', f.shape, f[0].sum())


if __name__ == '__main__':
    main()
