"""Per-shape timing of vtx_pool_conv_ln fwd / bwd on the MViT-B pooling shapes (32 clips, 8x56x56 token grid after the stem).
Select the library with VTX_LIB for an A/B of two builds on one box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch

import vtx
from vtx import functions as F_

DEV = torch.device('cuda', 0)
# (name, T, H, W, heads, stride)
SHAPES = [('blk0 kv  56x56 C96  s8', 8, 56, 56, 1, (1, 8, 8)), ('blk1 q   56x56 C192 s2', 8, 56, 56, 2, (1, 2, 2)),
          ('blk1 kv  56x56 C192 s8', 8, 56, 56, 2, (1, 8, 8)), ('blk2 kv  28x28 C192 s4', 8, 28, 28, 2, (1, 4, 4)),
          ('blk3 q   28x28 C384 s2', 8, 28, 28, 4, (1, 2, 2)), ('blk3 kv  28x28 C384 s4', 8, 28, 28, 4, (1, 4, 4)),
          ('blk4 kv  14x14 C384 s2', 8, 14, 14, 4, (1, 2, 2)), ('blk14 q  14x14 C768 s2', 8, 14, 14, 8, (1, 2, 2)),
          ('blk15 kv  7x7  C768 s1', 8, 7, 7, 8, (1, 1, 1))]


def main(B=32):
    print('lib:', os.environ.get('VTX_LIB', 'libvtx.so'))
    for name, T, H, W, heads, stride in SHAPES:
        C = heads * 96
        x = torch.randn(B, 1 + T * H * W, C, device=DEV, dtype=torch.bfloat16, requires_grad=True)
        w = torch.randn(96, 1, 3, 3, 3, device=DEV, requires_grad=True)
        g = torch.ones(96, device=DEV, requires_grad=True)
        b = torch.zeros(96, device=DEV, requires_grad=True)
        y = F_.PoolConvLNFn.apply(x, w, g, b, [T, H, W], heads, stride, 1e-5)
        dy = torch.randn_like(y)
        with torch.no_grad():
            f = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            f[0].record()
            for _ in range(5):
                F_.PoolConvLNFn.apply(x, w, g, b, [T, H, W], heads, stride, 1e-5)
            f[1].record()
        for _ in range(2):
            y.backward(dy, retain_graph=True)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(5):
            y.backward(dy, retain_graph=True)
        e[1].record()
        torch.cuda.synchronize()
        print(f'{name}: fwd {f[0].elapsed_time(f[1]) / 5 * 1e3:8.1f} us  bwd {e[0].elapsed_time(e[1]) / 5 * 1e3:8.1f} us', flush=True)
        del x, y, dy


if __name__ == '__main__':
    main()
