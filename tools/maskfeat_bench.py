"""MaskFeat pre-training step (BASELINE.json configs[3] on one GPU): MViT-B backbone + decoder + HOG-target masked MSE,
16x3x224x224 clips, fwd + bwd + fused AdamW.  GPU box only.

    python tools/maskfeat_bench.py [clips] [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vtx  # noqa: E402
import video_transformer as V  # noqa: E402
from vtx import ops, optim  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    vtx.set_precision('bf16')
    dev = 'cuda:0'
    torch.manual_seed(0)
    m = V.MaskFeat(pool_q_stride_size=[[1, 1, 2, 2], [3, 1, 2, 2]], feature_dim=216).to(dev).train()
    opt = optim.FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.05, clip_grad=0.02)
    x = torch.randn(B, 16, 3, 224, 224).to(dev)
    frames = torch.randint(0, 256, (B * 3, 224, 224, 3), dtype=torch.uint8, device=dev)      # centre frames of 3 cubes per clip
    mask = torch.zeros(B, 8, 14, 14, dtype=torch.int32)
    mask[:, 2:4, 3:9, 2:10] = 1
    mask[:, 6, 5:12, 5:12] = 1
    mask = mask.to(dev)
    markers = [[[2, 2], [6, 1]]] * B

    def step():
        hog = ops.hog_fwd(frames)                                   # targets on the device (dataset.py:188-196 does this on CPU workers)
        target = torch.zeros(B, 16, 14, 14, 108, dtype=torch.float64, device=dev)
        target[:, 6] = hog[0::3]
        target[:, 13] = hog[1::3]
        m.zero_grad(set_to_none=True)
        _, loss = m(x, target, mask, markers)
        loss.backward()
        opt.step()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'MaskFeat (MViT-B, 36.3 M params) bf16, {B} clips of 16x224^2: {dt * 1e3:.1f} ms/step = {B / dt:.1f} clips/s '
          f'(~{B / dt * 0.212:.1f} model TFLOP/s at 70.6 GFLOP fwd per clip), loss {float(loss):.4f}')


if __name__ == '__main__':
    main()
