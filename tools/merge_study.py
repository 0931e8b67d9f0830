"""bf16 margin study (GPU box only -- the reference values come from the CPU oracle, oracle/vt_oracle.py).

(a) vivit_small fact_encoder: the per-tensor gradient errors of the bf16 path behind the one number the parity report prints
    (worst l2-rel 1.74e-2 where the reference's autocast run has 1.02e-2): the tensors, sorted, next to the reference's own
    autocast deviation on each (tests/golden/autocast_cal.json), and the same with the fp32-VALU attention kernels (attn_valu=1)
    to separate the attention kernels from the GEMMs.
(b) the merged attn.proj o temporal_fc GEMM against the two GEMMs of the reference AT FULL SIZE: TimeSformer-B
    8x224^2, one clip, train mode, twelve weight seeds -- output error, worst and median parameter-gradient error of the bf16 path
    against the fp32 oracle, with set_merge_temporal_fc(True / False); and the step time of both settings.

    python tools/merge_study.py [a] [b] [--seeds 12]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

DEV = 'cuda:0'


def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).norm().item() / max(b.norm().item(), 1e-30)


def part_a():
    import vtx
    import video_transformer as V
    from oracle import synth
    from helpers import gold, NS
    from model_common import SMALL
    g = gold('vivit_small_fact_encoder.npz')
    cal = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'autocast_cal.json')))['vivit_small fact_encoder train']['grad']
    for valu in ('0', '1'):
        vtx.set_option('attn_valu', valu)
        vtx.set_precision('bf16')
        m = V.ViViT(num_frames=8, attention_type='fact_encoder', **SMALL)
        m.load_state_dict(synth.synth_state_dict(synth.shapes_of(m), 4), strict=True)
        m.to(DEV).train()
        torch.manual_seed(13)
        y = m(synth.synth_clip(3, 8, 3, 64, 64, seed=5).to(DEV))
        w = (synth.synth_tensor('loss_w', (128,), 0) * 10.0).to(DEV)
        (y * w).sum().backward()
        rows = []
        for k in g.files:
            if k.startswith('g:'):
                name = k[2:]
                rows.append((l2(dict(m.named_parameters())[name].grad, torch.as_tensor(g[k])), name))
            elif k.startswith('gs:') or k.startswith('gh:'):
                # large tensors are stored as 4096 strided samples ('gs:') or as their first 256 elements ('gh:'); the test's
                # metric for them is the larger of the part's relative L2 error (against the part's norm or the norm that many
                # rms-sized elements would have) and the error of the whole tensor's norm -- tests/helpers.py::compare_grads
                name = k[3:]
                full = dict(m.named_parameters())[name].grad.detach().double().cpu()
                flat = full.flatten()
                gn = g['gn:' + name]
                part = flat[:256] if k.startswith('gh:') else flat[torch.arange(0, min(NS, flat.numel())) * max(flat.numel() // NS, 1)]
                ref = torch.as_tensor(g[k]).double()
                rms = gn[0] / (full.numel() ** 0.5)
                e_part = (part - ref).norm().item() / max(ref.norm().item(), rms * ref.numel() ** 0.5)
                e_norm = abs(full.norm().item() - gn[0]) / max(gn[0], 1e-30)
                rows.append((max(e_part, e_norm), name + (' [first 256 elements]' if k.startswith('gh:') else ' [4096 samples]')))
        rows.sort(reverse=True)
        print(f'(a) vivit_small fact_encoder bf16, attn_valu={valu}: {len(rows)} gradients, median {rows[len(rows) // 2][0]:.3e}; the ten largest:')
        for e, name in rows[:10]:
            base = name.split(' [')[0]
            print(f'      {e:.3e}  (reference autocast {cal.get(base, float("nan")):.3e}; x{e / max(cal.get(base, 1e-30), 1e-30):.2f})  {name}')
    vtx.set_option('attn_valu', '0')


def part_b(nseeds):
    import vtx
    import video_transformer as V
    from vtx import functions as F_
    from oracle import synth, vt_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    x = synth.synth_clip(1, 8, seed=1)
    w = synth.synth_tensor('loss_w', (768,), 0) * 10.0
    shapes = synth.shapes_of(V.TimeSformer(num_frames=8))
    print(f'(b) TimeSformer-B 8x224^2, one clip, train mode (torch.manual_seed(7)), bf16 path vs the fp32 CPU oracle, {nseeds} weight seeds')
    print(f'{"seed":>4s} | {"merged: out":>11s} {"worst grad":>10s} {"median":>9s} | {"two GEMMs: out":>14s} {"worst grad":>10s} {"median":>9s} | worst tensor (merged / two GEMMs)')
    acc = {True: [], False: []}
    for seed in range(nseeds):
        sd = synth.synth_state_dict(shapes, seed)
        ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        torch.manual_seed(7)
        yo = O.timesformer_forward(ps, x, 8, training=True)
        (yo * w).sum().backward()
        res = {}
        for merged in (True, False):
            F_.set_merge_temporal_fc(merged)
            F_.clear_weight_cache()
            vtx.set_precision('bf16')
            m = V.TimeSformer(num_frames=8)
            m.load_state_dict(sd, strict=True)
            m.to(DEV).train()
            torch.manual_seed(7)
            y = m(x.to(DEV))
            (y * w.to(DEV)).sum().backward()
            eo = (y.detach().cpu().double() - yo.detach().double()).abs().max().item() / yo.detach().abs().max().item()
            errs = sorted((l2(p.grad, ps[k].grad), k) for k, p in m.named_parameters() if p.grad is not None)
            res[merged] = (eo, errs[-1][0], errs[len(errs) // 2][0], errs[-1][1])
            acc[merged].append(res[merged][:3])
            del m
        a, b = res[True], res[False]
        print(f'{seed:4d} | {a[0]:11.3e} {a[1]:10.3e} {a[2]:9.3e} | {b[0]:14.3e} {b[1]:10.3e} {b[2]:9.3e} | {a[3]} / {b[3]}', flush=True)
    for merged in (True, False):
        v = np.array(acc[merged])
        print(f'{"merged" if merged else "two GEMMs":>10s}: out mean {v[:, 0].mean():.3e} max {v[:, 0].max():.3e}; worst gradient mean {v[:, 1].mean():.3e} max {v[:, 1].max():.3e}; '
              f'median gradient mean {v[:, 2].mean():.3e}')
    # step time of both settings: fwd + bwd of 32 clips (no optimizer: the weights stay, the merged weight is re-formed per step
    # as in training because clear_weight_cache() runs between steps)
    xb = torch.randn(32, 8, 3, 224, 224, device=DEV)
    m = V.TimeSformer(num_frames=8).to(DEV).train()
    with torch.no_grad():
        for blk in m.transformer_layers.layers:
            blk.attentions[0].temporal_fc.weight.normal_(0, 0.02)
    for merged in (True, False, True, False):
        F_.set_merge_temporal_fc(merged)
        ts = []
        for i in range(6):
            F_.clear_weight_cache()
            m.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m(xb).sum().backward()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f'    fwd + bwd of 32 clips, merge_temporal_fc={merged}: {np.median(ts[2:]) * 1e3:.2f} ms')
    F_.set_merge_temporal_fc(True)


if __name__ == '__main__':
    which = [a for a in sys.argv[1:] if a in ('a', 'b')] or ['a', 'b']
    n = int(sys.argv[sys.argv.index('--seeds') + 1]) if '--seeds' in sys.argv else 12
    if 'a' in which:
        part_a()
    if 'b' in which:
        part_b(n)
