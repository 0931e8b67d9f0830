"""vtx_wprod on the three 768^3 products of the merged temporal projection (forward W_tfc W_proj + bias vector, backward
dW_tfc = c G W_proj^T + u b^T, dW_proj = c W_tfc^T G + vector): time per launch and a checksum of the result bits (two builds of the
library that sum in the same order print the same checksums).  VTX_LIB selects the build.  GPU box only."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
from vtx import ops
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(1)
D = 768
Wt, Wp, G = (torch.randn(D, D, generator=g, device=dev) * 0.03 for _ in range(3))
bp, bt, u = (torch.randn(D, generator=g, device=dev) for _ in range(3))
cases = {
    'fwd  W_tfc W_proj (+ W_tfc b_proj + b_tfc / c)': lambda: ops.wprod(Wt, Wp, x=bp, z=bt, beta_z=0.9),
    'bwd  c G W_proj^T + u b_proj^T': lambda: ops.wprod(G, Wp, tb=True, alpha=1.1, u=u, v=bp),
    'bwd  c W_tfc^T G (+ W_tfc^T u)': lambda: ops.wprod(Wt, G, ta=True, alpha=1.1, x=u),
}
for name, fn in cases.items():
    for _ in range(3):
        r = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    outs = r if isinstance(r, tuple) else (r,)
    h = hashlib.sha256(b''.join(t.detach().cpu().numpy().tobytes() for t in outs if t is not None)).hexdigest()[:16]
    print(f'{os.path.basename(os.environ.get("VTX_LIB", "libvtx.so")):18s} {name:48s} {e0.elapsed_time(e1) * 20:7.1f} us  bits {h}', flush=True)
