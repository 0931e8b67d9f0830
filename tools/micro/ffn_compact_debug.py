"""Where do the compact and the full FFN paths part?  Intermediate by intermediate, kept rows only (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch
import vtx
from vtx import ops, functions as F_
from vtx._lib import IDENT

DEV = 'cuda:0'
B, rows_per, D, Hd = 6, 1569, 768, 3072
dt = torch.bfloat16
g = torch.Generator().manual_seed(11)
r = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc)
x = r(B, rows_per, D).to(dt).to(DEV)
dout = r(B, rows_per, D).to(dt).to(DEV)
ln_w, ln_b = (1 + 0.1 * r(D)).to(DEV), (0.1 * r(D)).to(DEV)
w1, b1 = r(Hd, D, sc=D ** -0.5).to(DEV), (0.1 * r(Hd)).to(DEV)
w2, b2 = r(D, Hd, sc=Hd ** -0.5).to(DEV), (0.1 * r(D)).to(DEV)
c = float(np.float32(1.0) / np.float32(0.9))
dropped = [int(a) for a in sys.argv[1:]] or [5]
kept = [i for i in range(B) if i not in dropped]
host = torch.tensor([0.0 if i in dropped else c for i in range(B)], dtype=torch.float32)
sv = host.to(DEV)
w1c, w1T = F_.weights(torch.nn.Parameter(w1), dt, True)
w2c, w2T = F_.weights(torch.nn.Parameter(w2), dt, True)


def run(compact):
    out = {}
    M = B * rows_per
    if compact:
        sv._vtx_host = host
        plan = F_._compaction_plan(sv, B, rows_per, x.device)
        nk, nd = plan[0], plan[1]
        kmap, dmap, sv_k = F_._plan_maps(plan, rows_per)
        Mk = nk * rows_per
        xmap, smap, scale, rows = kmap, kmap, sv_k, Mk
    else:
        xmap, smap, scale, rows = IDENT, IDENT, sv, M
    e = lambda *s, d=dt: torch.empty(*s, dtype=d, device=DEV)
    xn, mean, rstd = e(rows, D), e(rows, d=torch.float32), e(rows, d=torch.float32)
    ops.layernorm_fwd(x, rows, D, D, xmap, ln_w, ln_b, 1e-5, xn, D, IDENT, mean, rstd)
    h, gg = e(rows, Hd), e(rows, Hd)
    ops.gemm_nt(xn, w1c, gg, rows, Hd, D, bias=b1, act=2, C2=h)
    y = torch.empty_like(x)
    ops.gemm_nt(gg, w2c, y, rows, D, Hd, cmap=xmap, bias=b2, row_scale=scale, rs=(rows_per, 1, 1, 0), R=x, rmap=xmap)
    dz = e(rows, D)
    ops.row_scale_copy(dout, dz, rows, D, smap=smap, s=scale, rs=(rows_per, 1, 1, 0))
    dh = e(rows, Hd)
    ops.gemm_nt(dz, w2T, dh, rows, Hd, D, dgelu_in=h, dgelu_kind=1)
    dxn = e(rows, D)
    ops.gemm_nt(dh, w1T, dxn, rows, D, Hd)
    dx = torch.empty_like(x)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    ops.layernorm_bwd(dxn, D, IDENT, x, D, xmap, rows, D, mean, rstd, ln_w, dout, dx, D, dg, db)
    torch.cuda.synchronize()
    pick = (lambda t: t) if compact else (lambda t: t.view(B, rows_per, -1)[kept].reshape(len(kept) * rows_per, -1))
    for name, t in (('xn', xn), ('mean', mean.view(-1, 1)), ('rstd', rstd.view(-1, 1)), ('h', h), ('g', gg), ('dz', dz), ('dh', dh), ('dxn', dxn)):
        out[name] = pick(t).float().cpu()
    out['y'] = y[kept].float().cpu()
    out['dx'] = dx[kept].float().cpu()
    return out


a, b_, b2_ = run(False), run(True), run(True)
for k in a:
    d = (a[k] - b_[k]).abs()
    nz = int((d > 0).sum())
    rows_bad = sorted(set((d.reshape(len(kept) * rows_per, -1) > 0).any(1).nonzero().flatten().tolist()))
    print(f'{k:5s} differing elements {nz:8d} max {d.max().item():.3e}  rows {rows_bad[:6]}{"..." if len(rows_bad) > 6 else ""} (n={len(rows_bad)}); '
          f'compact deterministic: {torch.equal(b_[k], b2_[k])}')
