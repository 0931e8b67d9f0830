"""Per-step wall times of the training step at several batch sizes (GPU box diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch
import vtx, transformer as T, video_transformer as V
dev = torch.device('cuda:0')
vtx.set_precision('bf16')
model = V.TimeSformer(num_frames=8).to(dev).train()
head = T.ClassificationHead(400, 768).to(dev).train()
mode = os.environ.get('DP_MODE')
if mode:
    import types
    def sv(self, rows, ndim, device):
        p = self.dropout_p
        if not p or not self.training: return None
        keep = 1 - p
        if mode == 'gpu_rng':
            return ((keep + torch.rand(rows, device=device)).floor_() / keep)
        if mode == 'cpu_rng_noupload':
            u = torch.rand((rows,) + (1,) * (ndim - 1))
            return torch.ones(rows, device=device)
        if mode == 'upload_only':
            return vtx.ops.upload_f32(torch.ones(rows), device)
    for m in model.modules():
        if isinstance(m, T.DropPath): m.scale_vector = types.MethodType(sv, m)
print('threads', torch.get_num_threads(), 'affinity', len(os.sched_getaffinity(0)), 'cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a', flush=True)
if os.environ.get('NO_DROPPATH'):
    for m in model.modules():
        if isinstance(m, T.DropPath): m.dropout_p = 0.0
params = list(model.parameters()) + list(head.parameters())
opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9, nesterov=True)
for B in [int(a) for a in sys.argv[1:]] or [16, 32, 48, 64]:
    x = torch.randn(B, 8, 3, 224, 224, device=dev)
    y = torch.randint(0, 400, (B,), device=dev)
    def step():
        for p in params: p.grad = None
        loss = torch.nn.functional.cross_entropy(head(model(x)), y)
        loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); step(); th = time.perf_counter() - t0
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0, th))
    t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / 8
    st = torch.cuda.memory_stats()
    print(f'B={B}: per-step wall ms {[round(a*1e3,1) for a,_ in ts]} host ms {[round(b*1e3,1) for _,b in ts]} | back-to-back {tb*1e3:.1f} ms/step = {B/tb:.1f} clips/s | '
          f'alloc_retries {st.get("num_alloc_retries")} reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB segments {st.get("segment.all.current")}', flush=True)
