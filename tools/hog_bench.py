"""HOG kernel rate: 1024 random uint8 frames 224x224x3 per launch (the worst case for the magnitude path: uniform random
pixels make every gradient pair equally likely).  VTX_LIB selects a diagnostic build (csrc/build.py --variant NAME HOG_ABLATE=n)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
from vtx import ops
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fr = torch.randint(0, 256, (F, 224, 224, 3), dtype=torch.uint8, device='cuda:0')
for _ in range(3):
    ops.hog_fwd(fr)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.hog_fwd(fr)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
gb = F * (150528 + 169344) / 1e9
print(f'{os.path.basename(os.environ.get("VTX_LIB", "libvtx.so"))}: {us:8.1f} us per {F} frames  {gb / us * 1e6:7.1f} GB/s = {gb / us * 1e6 / 8000:.3f} of 8 TB/s', flush=True)
