"""cProfile of the host side of the training step (GPU box): where does Python time go?"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import vtx  # noqa: E402
import transformer as T  # noqa: E402
import video_transformer as V  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
vtx.set_precision('bf16')
model = V.TimeSformer(num_frames=8).to(dev).train()
head = T.ClassificationHead(400, 768).to(dev).train()
params = list(model.parameters()) + list(head.parameters())
from vtx import dp, optim, functions as F_  # noqa: E402
buckets = dp.GradBuckets(params, direct=True)            # the bench.py stack
opt = optim.FusedSGD(buckets, lr=1e-4, momentum=0.9, nesterov=True)
x = torch.randn(B, 8, 3, 224, 224, device=dev)
y = torch.randint(0, 400, (B,), device=dev)


def step():
    buckets.zero()
    loss = F_.SoftmaxXentFn.apply(head(model(x)), y)
    loss.backward()
    buckets.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
# host-only time: enqueue without waiting
t0 = time.perf_counter()
for _ in range(5):
    step()
t_host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 5
print(f'B={B}: host enqueue {t_host*1e3:.1f} ms/step, wall {t_all*1e3:.1f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(35)
