"""Find host-side stalls: time every libvtx call and every torch.empty in the step; report outliers."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import vtx, transformer as T, video_transformer as V
from vtx import _lib, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2 and sys.argv[2] == 'nogc':
    gc.disable()
dev = torch.device('cuda:0')
vtx.set_precision('bf16')
model = V.TimeSformer(num_frames=8).to(dev).train()
head = T.ClassificationHead(400, 768).to(dev).train()
params = list(model.parameters()) + list(head.parameters())
opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9, nesterov=True)
x = torch.randn(B, 8, 3, 224, 224, device=dev)
y = torch.randint(0, 400, (B,), device=dev)
slow = []
orig_call = _lib.call
def timed_call(name, *a):
    t0 = time.perf_counter(); orig_call(name, *a); dt = time.perf_counter() - t0
    if dt > 2e-3: slow.append((name, round(dt * 1e3, 1)))
_lib.call = timed_call; ops.call = timed_call
orig_empty = torch.empty
def timed_empty(*a, **k):
    t0 = time.perf_counter(); r = orig_empty(*a, **k); dt = time.perf_counter() - t0
    if dt > 2e-3: slow.append(('torch.empty' + str(a[:1]) + str(k.get('pin_memory', '')), round(dt * 1e3, 1)))
    return r
torch.empty = timed_empty
def step():
    for p in params: p.grad = None
    t0 = time.perf_counter(); out = head(model(x)); t1 = time.perf_counter()
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward(); t2 = time.perf_counter(); opt.step(); t3 = time.perf_counter()
    return round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1), round((t3 - t2) * 1e3, 1)
for _ in range(3): step()
torch.cuda.synchronize(); slow.clear()
for i in range(8):
    t0 = time.perf_counter(); ph = step(); th = time.perf_counter() - t0
    print(f'step {i}: host {th*1e3:.1f} ms (fwd, bwd, opt) = {ph}; slow calls: {slow}', flush=True); slow.clear()
torch.cuda.synchronize()
