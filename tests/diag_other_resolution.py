"""Diagnostic (not collected by pytest): how far is the bf16 path from the fp32 oracle on the small other-resolution cases,
seed by seed, next to the REFERENCE's own torch.autocast(bfloat16) deviation on the same cases?

    python tests/diag_other_resolution.py hip  [n_seeds]     GPU box: this library, default options and a few variants
    python tests/diag_other_resolution.py ref  [n_seeds]     dev container: the unmodified reference, fp32 vs autocast

VERDICT r4: `test_timesformer_other_resolution_vs_oracle[(64, 96)-bf16]` measured 1.907e-2 against a fixed 1.5e-2 bar on
the driver's box; the reference's own autocast run deviates by 1.33e-2 on that case.  The metric is a maximum over 256
output values of a two-layer model, so it moves a lot from seed to seed: this script prints the distribution instead of
one draw (seed 6 is the test's).  VTX_LIB selects another build of the library for A/B runs.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from oracle import synth, vt_oracle as O  # noqa: E402

SMALL = dict(img_size=64, patch_size=16, embed_dims=128, num_heads=2, num_transformer_layers=2)
HWS = [(96, 96), (64, 96), (32, 32)]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()


def seeds(n):
    return [6] + list(range(100, 100 + n - 1))


def hip(n):
    import vtx
    import video_transformer as V
    dev = 'cuda:0'
    variants = [('default', {}, True), ('two GEMMs (merge off)', {}, False), ('ln_rows=1', {'ln_rows': '1'}, True),
                ('attn_hw_fwd=1', {'attn_hw_fwd': '1'}, True)]
    print('# lib:', os.environ.get('VTX_LIB', 'libvtx.so'))
    for hw in HWS:
        for seed in seeds(n):
            m = V.TimeSformer(num_frames=2, **SMALL)
            sd = synth.synth_state_dict(synth.shapes_of(m), seed)
            m.load_state_dict(sd, strict=True)
            m.to(dev).eval()
            x = synth.synth_clip(2, 2, 3, hw[0], hw[1], seed=3)
            with torch.no_grad():
                yo = O.timesformer_forward({k: v.clone() for k, v in sd.items()}, x, 2, heads=2, layers=2)
            row = []
            for name, opts, merge in variants:
                for k, v in opts.items():
                    vtx.set_option(k, v)
                vtx.functions.set_merge_temporal_fc(merge)
                vtx.functions.clear_weight_cache()
                vtx.set_precision('bf16')
                with torch.no_grad():
                    y = m(x.to(dev))
                row.append(rel(y, yo))
                vtx.set_precision('fp32')
                for k in opts:
                    vtx.set_option(k, {'ln_rows': '3', 'attn_hw_fwd': '16'}[k])
                vtx.functions.set_merge_temporal_fc(True)
            with torch.no_grad():
                y32 = m(x.to(dev))
            print(f'hip {hw} seed {seed:3d}: ' + '  '.join(f'{nm} {e:.3e}' for (nm, _, _), e in zip(variants, row)) +
                  f'  fp32 {rel(y32, yo):.1e}', flush=True)
    vtx.set_precision('auto')


def ref(n):
    from oracle import ref_loader
    VT = ref_loader.load().video_transformer
    for hw in HWS:
        for seed in seeds(n):
            m = VT.TimeSformer(num_frames=2, **SMALL)
            sd = synth.synth_state_dict(synth.shapes_of(m), seed)
            m.load_state_dict(sd, strict=True)
            m.eval()
            x = synth.synth_clip(2, 2, 3, hw[0], hw[1], seed=3)
            with torch.no_grad():
                y = m(x)
                with torch.autocast('cpu', dtype=torch.bfloat16):
                    ya = m(x).float()
            print(f'ref {hw} seed {seed:3d}: reference autocast {rel(ya, y):.3e}', flush=True)


if __name__ == '__main__':
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    {'hip': hip, 'ref': ref}[sys.argv[1]](n)
