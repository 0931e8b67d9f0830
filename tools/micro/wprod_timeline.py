"""Timeline of the wprod kernel (VTX_LIB = a -DVTX_WP_TRACE build: csrc/build.py --variant wptrace VTX_WP_TRACE=1).  Per wave six
stamps of the 100 MHz clock: 0 entry, 1 look-ahead loads issued, 2 first trip of the K loop done, 3 K loop done, 4 behind the fold
barrier, 5 exit; shader-clock counter at 0 and 3.  Prints segment means / p90 over all waves and the kernel's span."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import vtx
from vtx import ops
dev = 'cuda:0'
D = 768
g = torch.Generator(device=dev).manual_seed(1)
Wt, Wp, G = (torch.randn(D, D, generator=g, device=dev) * 0.03 for _ in range(3))
bp, bt, u = (torch.randn(D, generator=g, device=dev) for _ in range(3))
cases = {'fwd <true,false>': lambda: ops.wprod(Wt, Wp, x=bp, z=bt, beta_z=0.9),
         'bwd <true,true>': lambda: ops.wprod(G, Wp, tb=True, alpha=1.1, u=u, v=bp),
         'bwd <false,false>': lambda: ops.wprod(Wt, G, ta=True, alpha=1.1, x=u)}
nw = 24 * 24 * 4
for name, fn in cases.items():
    for _ in range(3):
        fn()
    trace = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
    vtx.set_option('pp_trace', str(trace.data_ptr()))
    fn()
    torch.cuda.synchronize()
    vtx.set_option('pp_trace', '0')
    t = trace.cpu().view(nw, 8).double()
    t0 = t[:, 0].min()
    span = (t[:, 5].max() - t0) / 100.0
    print(f'{name}: kernel span (first entry -> last exit) {span:.1f} us; wave entry spread {((t[:, 0].max() - t0) / 100):.1f} us')
    for a, b, what in ((0, 1, 'entry -> look-ahead issued'), (1, 2, 'first trip (first loads land + 16 MFMAs)'), (2, 3, 'rest of the K loop'),
                       (3, 4, 'fold barrier'), (4, 5, 'fold + store')):
        d = (t[:, b] - t[:, a]) / 100.0
        print(f'   {what:42s} mean {d.mean():6.2f} us  p90 {d.quantile(0.9):6.2f}  max {d.max():6.2f}')
    life = (t[:, 5] - t[:, 0]) / 100.0
    cyc = (t[:, 7] - t[:, 6])
    print(f'   wave lifetime mean {life.mean():.2f} us; shader cycles entry -> loop end mean {cyc.mean():.0f} (= {cyc.mean() / ((t[:, 3] - t[:, 0]).mean() * 10):.2f} GHz)')
