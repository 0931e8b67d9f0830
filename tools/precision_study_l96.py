"""Why the bf16 path's output deviation grows with depth (dev container only: needs /root/reference).

The UNMODIFIED reference TimeSformer-L (D 1024, 16 heads, 24 layers) on one 96x3x224x224 clip, eval-mode forward on the CPU:
its own torch.autocast(bfloat16) run keeps the residual stream in float32 (autocast rounds the matmul operands only); the HIP
path stores the stream as bf16.  This tool adds exactly that -- a round-to-bf16 of every sub-block's output
(3 per layer: temporal attention, spatial attention, FFN) -- to the reference's autocast run and prints the output deviation of
both from the reference's fp32 run (tests/golden/tsf_l_t96_d24_eval.npz), for the first `depth` layers of the same weights.

    python tools/precision_study_l96.py [depth ...]            (default 24)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader, synth  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count())
    depths = [int(a) for a in sys.argv[1:]] or [24]
    VT = ref_loader.load().video_transformer
    x = synth.synth_clip(1, 96, seed=5)
    for depth in depths:
        m = VT.TimeSformer(num_frames=96, embed_dims=1024, num_heads=16, num_transformer_layers=depth)
        m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), seed=0), strict=True)
        m.eval()
        with torch.no_grad():
            if depth == 24:
                y0 = torch.from_numpy(np.load(os.path.join(ROOT, 'tests', 'golden', 'tsf_l_t96_d24_eval.npz'))['out'])
            else:
                y0 = m(x).float()
            with torch.autocast('cpu', dtype=torch.bfloat16):
                ya = m(x).float()
            hooks = []
            for layer in m.transformer_layers.layers:
                for sub in list(layer.attentions) + list(layer.ffns):
                    hooks.append(sub.register_forward_hook(lambda mod, inp, out: out.float().bfloat16().float() if torch.is_tensor(out) else out))
            with torch.autocast('cpu', dtype=torch.bfloat16):
                ys = m(x).float()
            for h in hooks:
                h.remove()
        dev = lambda y: float((y - y0).abs().max() / y0.abs().max())      # noqa: E731
        print(f'depth {depth:2d}: reference autocast (fp32 stream) {dev(ya):.3e}   autocast + stream rounded to bf16 after every sub-block {dev(ys):.3e}', flush=True)


if __name__ == '__main__':
    main()
