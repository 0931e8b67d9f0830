"""Host->device time of one 64-clip batch: fp32 [B,T,C,H,W] (the reference's input) vs the decoded uint8
[B,T,H,W,C] clip that vtx_patch_rows_u8 consumes.  Pinned host memory, copy on the compute stream."""
import time
import torch
B, T = 64, 8
for name, t in (('fp32 [B,T,C,H,W]', torch.empty(B, T, 3, 224, 224, dtype=torch.float32).pin_memory()),
                ('uint8 [B,T,H,W,C]', torch.empty(B, T, 224, 224, 3, dtype=torch.uint8).pin_memory())):
    d = torch.empty_like(t, device='cuda')
    for _ in range(3):
        d.copy_(t, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        d.copy_(t, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f'{name}: {t.numel() * t.element_size() / 1e6:.1f} MB in {dt * 1e3:.2f} ms = {t.numel() * t.element_size() / dt / 1e9:.1f} GB/s')
