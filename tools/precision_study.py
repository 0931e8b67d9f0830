"""Where does the bf16 path's gradient error come from?  (dev container only: needs /root/reference)

Runs the UNMODIFIED reference TimeSformer-B (8x224^2, one clip, train mode) on the CPU in fp32 and
under torch.autocast(bfloat16), optionally rounding the residual stream to bf16 after every
sub-block in the forward pass, the backward pass, or both -- the storage choices open to the HIP
path.  Prints the worst / median relative-L2 parameter-gradient error of each variant against the
fp32 run, which is how the fp32 residual stream of the bf16 path was decided.

    python tools/precision_study.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader, synth  # noqa: E402


class RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def run(VT, sd, x, w, mode):
    m = VT.TimeSformer(num_frames=x.shape[1])
    m.load_state_dict(sd, strict=True)
    m.train(True)
    torch.manual_seed(7)
    hooks = []
    if mode not in ('fp32', 'autocast'):
        def hook(mod, inp, out):
            if not torch.is_tensor(out):
                return out
            out = out.float()
            if 'F' in mode:
                out = RoundFwd.apply(out)
            if 'B' in mode:
                out = RoundBwd.apply(out)
            return out
        for layer in m.transformer_layers.layers:
            for sub in list(layer.attentions) + list(layer.ffns):
                hooks.append(sub.register_forward_hook(hook))
    if mode == 'fp32':
        y = m(x)
    else:
        with torch.autocast('cpu', dtype=torch.bfloat16):
            y = m(x)
    (y.float() * w).sum().backward()
    for h in hooks:
        h.remove()
    return y.detach().float(), {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}


def main():
    torch.set_num_threads(os.cpu_count())
    VT = ref_loader.load().video_transformer
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    m0 = VT.TimeSformer(num_frames=T)
    sd = synth.synth_state_dict(synth.shapes_of(m0), seed=0)
    x = synth.synth_clip(1, T, seed=1)
    w = synth.synth_tensor('loss_w', (768,), 0) * 10.0
    y0, g0 = run(VT, sd, x, w, 'fp32')
    for mode in ('autocast', 'autocast+F', 'autocast+B', 'autocast+FB'):
        y, g = run(VT, sd, x, w, mode)
        eo = (y - y0).abs().max().item() / y0.abs().max().item()
        errs = sorted(((g[k] - g0[k]).norm().item() / max(g0[k].norm().item(), 1e-30), k) for k in g0)
        med = errs[len(errs) // 2][0]
        print(f'{mode:12s} out max-rel {eo:.3e}   grad l2-rel: worst {errs[-1][0]:.3e} ({errs[-1][1]})  '
              f'2nd {errs[-2][0]:.3e}  median {med:.3e}', flush=True)


if __name__ == '__main__':
    main()
