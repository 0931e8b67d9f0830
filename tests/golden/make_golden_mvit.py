"""Golden vectors of the full-size MaskFeat / MViT-B case (BASELINE cfg 4) from the CPU oracle (dev container or any host):

    python tests/golden/make_golden_mvit.py

  maskfeat_mvit_b_full.npz   oracle/mvit_cases.reference() in float64 on one 16x224^2 clip: 4096 strided samples + norm of
                             the prediction, the loss, every parameter gradient ('g:' whole / 'gs:' samples + 'gn:' norm,
                             sum; 'zero:' for gradients that vanish identically), and 'ae:' = per-parameter relative-L2 deviation of the SAME graph under
                             torch.autocast(bfloat16) from its float64 run -- the calibration of the bf16 bars.

PARITY UNPINNED: the backbone oracle restates an absent third-party package (oracle/mvit_oracle.py); this file makes the
full-size bf16 test of tests/test_gpu_mvit.py fast and its bar fixed, it does not pin the oracle."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import mvit_cases as MC  # noqa: E402

NS = 4096
SEED = 9


def sample_idx(numel):
    step = max(numel // NS, 1)
    return torch.arange(0, min(NS, numel)) * step


def main():
    torch.set_num_threads(os.cpu_count())
    oracle = MC.make_oracle()
    oracle.load_state_dict(MC.backbone_state(oracle, SEED), strict=True)
    head = MC.head_state(SEED)
    x, target, mask, markers = MC.inputs()
    t0 = time.time()
    pred, loss, g64 = MC.reference(oracle, head, x, target, mask, markers)
    print(f'float64 run {time.time() - t0:.0f} s, loss {loss.item():.6f}', flush=True)
    t0 = time.time()
    pa, la, gac = MC.reference(oracle, head, x, target, mask, markers, autocast=True)
    print(f'autocast run {time.time() - t0:.0f} s, loss {la.item():.6f}', flush=True)
    res = {'loss': np.array(loss.item()), 'loss_autocast': np.array(la.item())}
    pf = pred.flatten()
    res['pred_s'] = pf[sample_idx(pf.numel())].numpy()
    res['pred_n'] = np.array([pf.norm().item(), pf.abs().max().item()])
    res['pred_autocast_err'] = np.array((pa - pred).abs().max().item() / pred.abs().max().item())
    norms = sorted(g.norm().item() for g in g64.values())
    typical = norms[len(norms) // 2]
    res['typical_norm'] = np.array(typical)
    for k, g in g64.items():
        f = g.flatten()
        if f.norm().item() < 1e-6 * typical:        # exactly zero by symmetry (norm_k.bias: softmax ignores a common key shift)
            res['zero:' + k] = np.array(f.norm().item())
            continue
        if f.numel() <= 17000:
            res['g:' + k] = g.float().numpy()
        else:
            res['gs:' + k] = f[sample_idx(f.numel())].float().numpy()
            res['gn:' + k] = np.array([f.norm().item(), f.sum().item()])
        res['ae:' + k] = np.array((gac[k] - g).norm().item() / max(g.norm().item(), 1e-30))
    np.savez_compressed(os.path.join(HERE, 'maskfeat_mvit_b_full.npz'), **res)
    ae = sorted((float(v), k) for k, v in res.items() if k.startswith('ae:'))
    print('autocast pred err', float(res['pred_autocast_err']), 'grad l2 median', ae[len(ae) // 2], 'worst', ae[-5:])


if __name__ == '__main__':
    main()
