"""Throughput of the other BASELINE.json configurations' GPU pieces (not part of bench.py's contract line):
ViViT-B fact_encoder (Conv3d tubelets, 16 frames) fwd+bwd+SGD, and the MaskFeat HOG-target extractor with
its HBM roofline and the CPU C-oracle timed beside it.  Prints one JSON object per line."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch

import __graft_entry__ as ge
ge.ensure_built()
import vtx
from vtx import ops
import transformer as T
import video_transformer as V

DEV = torch.device('cuda', 0)


def vivit(batch=32, steps=8, warmup=3):
    vtx.set_precision('bf16')
    torch.manual_seed(0)
    m = V.ViViT(num_frames=16).to(DEV).train()
    head = T.ClassificationHead(400, m.embed_dims).to(DEV).train()
    params = list(m.parameters()) + list(head.parameters())
    opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9, nesterov=True)
    x = torch.randn(batch, 16, 3, 224, 224, device=DEV)
    y = torch.randint(0, 400, (batch,), device=DEV)

    def step():
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(head(m(x)), y)
        loss.backward()
        opt.step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flops = 0.850e12                      # ViViT-B fact_encoder T=16 fwd+bwd per clip
    print(json.dumps({'config': 'ViViT-B fact_encoder, Conv3d tubelet 2, 16x3x224x224, bf16, fwd+CE+bwd+SGD',
                      'clips_per_gpu': batch, 'clips_per_s': round(batch / dt, 2), 'ms_per_step': round(dt * 1e3, 2),
                      'model_tflops': round(batch / dt * flops / 1e12, 1),
                      'mfma_frac': round(batch / dt * flops / 1e12 / 2500.0, 4)}), flush=True)


def hog(frames_n=256, iters=20):
    frames = torch.randint(0, 256, (frames_n, 224, 224, 3), dtype=torch.uint8, device=DEV)
    for _ in range(3):
        ops.hog_fwd(frames)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.hog_fwd(frames)
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / iters
    bytes_per_frame = 150528 + 169344     # uint8 frame in, float64 [14,14,108] out
    res = {'config': 'MaskFeat HOG targets, 224x224x3 uint8 frames -> [14,14,108] float64',
           'frames_per_s': round(frames_n / dt, 0), 'achieved_GBps': round(frames_n * bytes_per_frame / dt / 1e9, 1),
           'hbm_frac': round(frames_n * bytes_per_frame / dt / 8e12, 4)}
    so = os.path.join(ROOT, 'oracle', '_build', 'libhogref.so')
    if os.path.isfile(so):                # the C oracle, one core, as the CPU baseline of this piece
        lib = ctypes.CDLL(so)
        f = np.random.RandomState(0).randint(0, 256, (224, 224, 3), dtype=np.uint8)
        out = np.empty((14, 14, 108), dtype=np.float64)
        fn = getattr(lib, 'vtx_ref_hog_frame', None)
        if fn is not None:
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            fn.restype = None
            t0 = time.perf_counter()
            for _ in range(5):
                fn(f.ctypes.data_as(ctypes.c_void_p), 224, 224, out.ctypes.data_as(ctypes.c_void_p), None)
            res['cpu_oracle_frames_per_s_1core'] = round(5 / (time.perf_counter() - t0), 1)
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    hog()
    vivit()
