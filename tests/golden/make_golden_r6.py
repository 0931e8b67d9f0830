"""Round-6 golden from the UNMODIFIED reference (dev container only: needs /root/reference):

    python tests/golden/make_golden_r6.py [tsf_l_t96_full] [autocast]

  tsf_l_t96_d24_eval.npz   BASELINE.json configs[4] at FULL depth: TimeSformer-L (D 1024, 16 heads, 24 layers) on one 96x3x224x224
                           clip, eval-mode forward -- 18 817 tokens through 24 layers (~17 TFLOP on the CPU).  VERDICT r5 item 7c:
                           the only TimeSformer-L golden so far was depth 2; 24 layers of bf16 residual stream is where accumulated
                           rounding would show.  'out' = the reference's fp32 features [1, 1024]; with `autocast` also
                           'out_autocast' = the reference's own torch.autocast(bfloat16) run (the yardstick for the bf16 path).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_loader, synth  # noqa: E402


def main():
    which = sys.argv[1:] or ['tsf_l_t96_full', 'autocast']
    VT = ref_loader.load().video_transformer
    torch.set_num_threads(os.cpu_count())
    path = os.path.join(HERE, 'tsf_l_t96_d24_eval.npz')
    if 'tsf_l_t96_full' in which:
        m = VT.TimeSformer(num_frames=96, embed_dims=1024, num_heads=16, num_transformer_layers=24)
        sd = synth.synth_state_dict(synth.shapes_of(m), seed=0)
        m.load_state_dict(sd, strict=True)
        m.eval()
        x = synth.synth_clip(1, 96, seed=5)
        res = {}
        t0 = time.time()
        with torch.no_grad():
            y = m(x).float()
        res['out'] = y.numpy()
        print('tsf_l_t96_d24_eval fp32', y.shape, f'{time.time() - t0:.0f} s', float(y.abs().max()), flush=True)
        np.savez_compressed(path, **res)
        if 'autocast' in which:
            t0 = time.time()
            with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
                yac = m(x).float()
            res['out_autocast'] = yac.numpy()
            print('tsf_l_t96_d24_eval autocast', f'{time.time() - t0:.0f} s', 'deviation', float((yac - y).abs().max() / y.abs().max()), flush=True)
            np.savez_compressed(path, **res)


if __name__ == '__main__':
    main()
