"""Diagnostic (GPU box, not collected by pytest): where does the bf16 path deviate from the fp32 path?

    python tests/diag_bf16_error_map.py [frames] > gpurun_out/bf16_error_map.txt

Runs TimeSformer-B (one clip, train mode, the seeds of tests/golden/tsf_b_t8_autocast.npz) through the
HIP path in fp32 (which matches the reference to ~1e-6, tests/test_gpu_models.py) and in bf16, and
prints, per variant: the relative L2 error of the residual stream after every sub-block (forward),
of the stream gradient entering every sub-block (backward), and of every parameter gradient (whole
tensor and the 256-element head the golden files store).  Variants toggle kernel families through
their environment switches so that a deviation can be pinned on one of them.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from oracle import synth  # noqa: E402

DEV = 'cuda:0'


def l2(a, b):
    a, b = a.double(), b.double()
    return (a - b).norm().item() / max(b.norm().item(), 1e-30)


def run(prec, frames, env):
    import vtx
    import video_transformer as V
    for k, v in env.items():
        vtx.set_option(k, v)
    vtx.set_precision(prec)
    vtx.functions.clear_weight_cache()
    m = V.TimeSformer(num_frames=frames)
    m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), 0), strict=True)
    m.to(DEV).train()
    fwd, bwd = {}, {}
    hooks = []
    for li, layer in enumerate(m.transformer_layers.layers):
        for name, sub in [('time', layer.attentions[0]), ('space', layer.attentions[1]), ('ffn', layer.ffns[0])]:
            tag = f'L{li:02d}.{name}'

            def hook(mod, inp, out, tag=tag):
                fwd[tag] = out.detach().float().cpu()
                out.register_hook(lambda g, tag=tag: bwd.__setitem__(tag, g.detach().float().cpu()))
            hooks.append(sub.register_forward_hook(hook))
    torch.manual_seed(7)
    y = m(synth.synth_clip(1, frames, seed=1).to(DEV))
    w = (synth.synth_tensor('loss_w', (768,), 0) * 10.0).to(DEV)
    (y * w).sum().backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.grad is not None}
    for k in env:
        vtx.set_option(k, 'auto' if k.startswith('gemm') else '0')
    return y.detach().float().cpu(), fwd, bwd, grads


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    y0, f0, b0, g0 = run('fp32', frames, {})
    variants = [('bf16 default', {}), ('bf16 attn_valu=1', {'attn_valu': '1'}),
                ('bf16 gemm_nt=dma2 gemm_tn=dma2', {'gemm_nt': 'dma2', 'gemm_tn': 'dma2'})]
    for name, env in variants:
        y, f, b, g = run('bf16', frames, env)
        print(f'==== {name}: out max-rel {(y - y0).abs().max().item() / y0.abs().max().item():.3e}  l2 {l2(y, y0):.3e}')
        print('  forward stream l2 after each sub-block / backward stream-gradient l2 at its output:')
        for tag in sorted(f0):
            print(f'    {tag:12s} fwd {l2(f[tag], f0[tag]):.3e}   bwd {l2(b[tag], b0[tag]):.3e}')
        rows = []
        for k in g0:
            full = l2(g[k], g0[k])
            h, h0 = g[k].flatten()[:256], g0[k].flatten()[:256]
            rms = g0[k].double().norm().item() / g0[k].numel() ** 0.5
            head = (h.double() - h0.double()).norm().item() / max(h0.double().norm().item(), rms * h0.numel() ** 0.5, 1e-30)
            rows.append((full, head, k))
        rows.sort(reverse=True)
        print('  parameter gradients, worst 25 by whole-tensor l2 (full, head-256 metric):')
        for full, head, k in rows[:25]:
            print(f'    {full:.3e}  {head:.3e}  {k}')
        print(f'  median full {rows[len(rows) // 2][0]:.3e}; worst head metric {max(r[1] for r in rows):.3e} '
              f'({max(rows, key=lambda r: r[1])[2]})', flush=True)


if __name__ == '__main__':
    main()
