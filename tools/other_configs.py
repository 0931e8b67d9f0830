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
from vtx import ops, functions as F_
import transformer as T
import video_transformer as V

DEV = torch.device('cuda', 0)


def train(name, model, batch, frames, flops_per_clip, steps=6, warmup=2, recompute=False):
    """fwd + cross-entropy + bwd + fused SGD-nesterov exactly as bench.py does it (gradient buckets with direct
    parameter gradients), bf16 path, synthetic clips resident in HBM."""
    from vtx import dp, optim
    vtx.set_precision('bf16')
    vtx.set_recompute(recompute)
    torch.manual_seed(0)
    torch.cuda.reset_peak_memory_stats()
    m = model.to(DEV).train()
    head = T.ClassificationHead(400, m.embed_dims).to(DEV).train()
    params = list(m.parameters()) + list(head.parameters())
    buckets = dp.GradBuckets(params, direct=True)
    opt = optim.FusedSGD(buckets, lr=1e-4, momentum=0.9, nesterov=True)
    x = torch.randn(batch, frames, 3, 224, 224, device=DEV)
    y = torch.randint(0, 400, (batch,), device=DEV)

    def step():
        buckets.zero()
        loss = F_.SoftmaxXentFn.apply(head(m(x)), y)
        loss.backward()
        buckets.finish()
        opt.step()
    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        peak0 = torch.cuda.max_memory_allocated()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res = {'config': name, 'clips_per_gpu': batch, 'clips_per_s': round(batch / dt, 2), 'ms_per_step': round(dt * 1e3, 2),
               'recompute': recompute, 'peak_GB': round(max(peak0, torch.cuda.max_memory_allocated()) / 2 ** 30, 1)}
        if flops_per_clip:
            res['model_tflops'] = round(batch / dt * flops_per_clip / 1e12, 1)
            res['mfma_frac'] = round(batch / dt * flops_per_clip / 1e12 / 2500.0, 4)
        print(json.dumps(res), flush=True)
    finally:
        buckets.remove()
        vtx.set_recompute(False)
        del m, head, params, opt, x
        torch.cuda.empty_cache()


def vivit(batch=32):
    # ViViT-B fact_encoder T=16 fwd+bwd per clip
    train('ViViT-B fact_encoder, Conv3d tubelet 2, 16x3x224x224, bf16, fwd+CE+bwd+SGD (BASELINE cfg 2)',
          V.ViViT(num_frames=16), batch, 16, 0.850e12)


def timesformer_b16(batch=48):
    train('TimeSformer-B divided_space_time, 16x3x224x224, bf16, fwd+CE+bwd+SGD', V.TimeSformer(num_frames=16), batch, 16, 2.352e12)


def timesformer_l96(batch=4, recompute=True):
    """BASELINE cfg 4: TimeSformer-L (D 1024, 16 heads, 24 layers), 96 frames: 18 817 tokens per clip.  Per-block
    recompute keeps one block's activations (2.1 GB per clip) instead of 24.  FLOPs per clip fwd+bwd (recompute not
    counted): 6 * tokens * 12 D^2 * 24 layers for the Linears + attention cores."""
    tokens, D, layers = 196 * 96 + 1, 1024, 24
    lin = 6.0 * tokens * (13 * D * D) * layers            # qkv 3 + proj 1 + temporal qkv 3 + proj 1 + tfc 1 + FFN 8 = 17 D^2? see below
    # per layer Linears: temporal qkv (3 D^2) + proj (D^2) + temporal_fc (D^2) + spatial qkv (3 D^2) + proj (D^2) + FFN (8 D^2) = 17 D^2
    lin = 6.0 * tokens * (17 * D * D) * layers
    attn = 3.0 * layers * (4.0 * tokens * 96 * D + 4.0 * tokens * 197 * D)     # fwd 4 L hd per token and head, x3 for fwd+bwd
    train('TimeSformer-L divided_space_time (D 1024, 24 layers), 96x3x224x224, bf16, fwd+CE+bwd+SGD, '
          + ('per-block recompute' if recompute else 'all activations stored (2.1 GB per layer and clip)') + ' (BASELINE cfg 5)',
          V.TimeSformer(num_frames=96, embed_dims=1024, num_heads=16, num_transformer_layers=24), batch, 96,
          lin + attn, steps=3, warmup=1, recompute=recompute)


def hog(frames_n=1024, iters=10):
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
    which = sys.argv[1:] or ['hog', 'vivit', 'tsf16', 'tsfl96']
    if 'hog' in which:
        hog()
    if 'vivit' in which:
        vivit()
    if 'tsf16' in which:
        timesformer_b16()
    if 'tsfl96' in which:
        timesformer_l96()
    if 'tsfl96_stored' in which:                      # 4 clips with every activation stored: ~205 GB of the 288
        timesformer_l96(batch=4, recompute=False)
    if 'tsfl96_12' in which:
        timesformer_l96(batch=12)
