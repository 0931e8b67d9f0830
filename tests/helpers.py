"""Shared test helpers: golden loading, error metrics, parity report."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
REPORT = os.path.join(ROOT, 'gpurun_out', 'parity_report.txt')

# parity bars (BASELINE.json north_star): fp32 path 1e-3 relative to the reference's fp32 CPU
# result; bf16 path is reported against the same fp32 oracle with the documented looser bar
# (a bf16 autocast CPU run of the reference itself deviates by 8.6e-3).
TOL_F32 = 1e-3
TOL_BF16 = 3e-2
TOL_BF16_GRAD = 6e-2


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def gold_keys():
    return json.load(open(os.path.join(GOLD, 'state_dict_keys.json')))


def relerr(a, b):
    """max|a-b| / max|b| in float64 (b = reference)."""
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    den = b.abs().max().item()
    return (a - b).abs().max().item() / max(den, 1e-30)


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, 'a') as f:
        f.write(line + '\n')


def check(name, got, ref, tol):
    e = relerr(got, ref)
    report(f'{"ok  " if e <= tol else "FAIL"} {name}: rel={e:.3e} (tol {tol:g})')
    assert e <= tol, f'{name}: rel err {e:.3e} > {tol:g}'
    return e


def compare_grads(prefix, named_grads, g, tol, tol_norm=None):
    """named_grads: {param_name: grad tensor}; g: golden npz from make_golden.grads_summary."""
    worst = 0.0
    n = 0
    for k in g.files:
        if k.startswith('g:'):
            e = relerr(named_grads[k[2:]], g[k])
        elif k.startswith('gh:'):
            name = k[3:]
            gn = g['gn:' + name]
            got = named_grads[name].detach().double().cpu()
            # head of the tensor relative to the tensor's rms, plus norm / sum checks
            rms = gn[0] / (got.numel() ** 0.5)
            e = (got.flatten()[:256] - torch.as_tensor(g[k]).double()).abs().max().item() / max(
                torch.as_tensor(g[k]).abs().max().item(), rms, 1e-30)
            e = max(e, abs(got.norm().item() - gn[0]) / max(gn[0], 1e-30))
        else:
            continue
        n += 1
        worst = max(worst, e)
        if e > tol:
            report(f'FAIL {prefix} grad {k}: rel={e:.3e}')
        assert e <= tol, f'{prefix}: grad {k} rel err {e:.3e} > {tol:g}'
    report(f'ok   {prefix}: {n} parameter gradients, worst rel={worst:.3e} (tol {tol:g})')
    return worst
