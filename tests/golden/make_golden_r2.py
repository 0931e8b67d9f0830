"""Round-2 golden vectors from the UNMODIFIED reference (dev container only):

    python tests/golden/make_golden_r2.py [tsf_b_t8_ac] [tsf_b_t16] [vivit_b_t16] [tsf_l_t96]

  tsf_b_t8_autocast.npz   TimeSformer-B 8x224^2 (clip seed 1, torch seed 7): the reference's own fp32 run AND its
                          own torch.autocast(bfloat16) run -- per-parameter relative-L2 deviation of the autocast
                          gradients from the fp32 ones ('ae:'), the calibration of the bf16 parity bar
  tsf_b_t16_train.npz     TimeSformer-B 16x224^2, batch 1, train mode fwd+bwd (the north_star's second clip shape)
  vivit_b_t16_train.npz   ViViT-B fact_encoder, Conv3d tubelets, 16x224^2, batch 2, train mode fwd+bwd (BASELINE cfg 3)
  tsf_l_t96_d2_train.npz  TimeSformer-L geometry (D 1024, 16 heads, 96 frames) at depth 2, batch 1, train fwd+bwd (cfg 5)

Gradients of large tensors are stored as 4096 evenly strided samples ('gs:') plus norm and sum ('gn:'), small ones whole
('g:').
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_loader, synth  # noqa: E402

NS = 4096


def sample_idx(numel):
    step = max(numel // NS, 1)
    return torch.arange(0, min(NS, numel)) * step


def grads_summary(model, full_limit=17000):
    out = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().double().flatten()
        if g.numel() <= full_limit:
            out['g:' + k] = g.reshape(p.shape).float().numpy()
        else:
            out['gs:' + k] = g[sample_idx(g.numel())].float().numpy()
            out['gn:' + k] = np.array([g.norm().item(), g.sum().item()])
    return out


def run(model, x, sd, seed, autocast=False):
    model.load_state_dict(sd, strict=True)
    model.train(True)
    model.zero_grad()
    torch.manual_seed(seed)
    if autocast:
        with torch.autocast('cpu', dtype=torch.bfloat16):
            y = model(x)
    else:
        y = model(x)
    w = synth.synth_tensor('loss_w', (y.shape[-1],), 0) * 10.0
    (y.float() * w).sum().backward()
    return y.detach().float()


def main():
    which = sys.argv[1:] or ['tsf_b_t8_ac', 'tsf_b_t16', 'vivit_b_t16', 'tsf_l_t96']
    VT = ref_loader.load().video_transformer
    torch.set_num_threads(os.cpu_count())
    if 'tsf_b_t8_ac' in which:
        m = VT.TimeSformer(num_frames=8)
        sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
        x = synth.synth_clip(1, 8, seed=1)
        y32 = run(m, x, sd, 7)
        res = grads_summary(m)
        g32 = {k: p.grad.detach().double().clone() for k, p in m.named_parameters()}
        yac = run(m, x, sd, 7, autocast=True)
        res['out'] = y32.numpy()
        res['out_autocast'] = yac.numpy()
        for k, p in m.named_parameters():
            res['ae:' + k] = np.array((p.grad.double() - g32[k]).norm().item() / max(g32[k].norm().item(), 1e-30))
        np.savez_compressed(os.path.join(HERE, 'tsf_b_t8_autocast.npz'), **res)
        ae = sorted(float(v) for k, v in res.items() if k.startswith('ae:'))
        print('tsf_b_t8_autocast: out dev', float((yac - y32).abs().max() / y32.abs().max()), 'grad l2 median/worst', ae[len(ae) // 2], ae[-1])
    if 'tsf_b_t16' in which:
        m = VT.TimeSformer(num_frames=16)
        sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
        y = run(m, synth.synth_clip(1, 16, seed=21), sd, 9)
        res = grads_summary(m)
        res['out'] = y.numpy()
        np.savez_compressed(os.path.join(HERE, 'tsf_b_t16_train.npz'), **res)
        print('tsf_b_t16_train', y.shape, len(res))
    if 'vivit_b_t16' in which:
        m = VT.ViViT(num_frames=16)
        sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
        y = run(m, synth.synth_clip(2, 16, seed=3), sd, 17)
        res = grads_summary(m)
        res['out'] = y.numpy()
        np.savez_compressed(os.path.join(HERE, 'vivit_b_t16_train.npz'), **res)
        print('vivit_b_t16_train', y.shape, len(res))
    if 'tsf_l_t96' in which:
        m = VT.TimeSformer(num_frames=96, embed_dims=1024, num_heads=16, num_transformer_layers=2)
        sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
        y = run(m, synth.synth_clip(1, 96, seed=5), sd, 19)
        res = grads_summary(m)
        res['out'] = y.numpy()
        np.savez_compressed(os.path.join(HERE, 'tsf_l_t96_d2_train.npz'), **res)
        print('tsf_l_t96_d2_train', y.shape, len(res))


if __name__ == '__main__':
    main()
