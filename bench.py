#!/usr/bin/env python
"""Headline benchmark: TimeSformer-B divided_space_time, 8 x 3 x 224 x 224 synthetic clips,
bf16 HIP path, one training step = forward + cross-entropy + backward (+ gradient
all-reduce for N > 1) + SGD update, on N GPUs of one node (one process per GPU, RCCL).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--frames T] [--stream bf16|fp32|fp32+grad]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torchrun environment re-launches itself under
torch.distributed.run with N ranks (one per GPU, rendezvous on 127.0.0.1); if the node has fewer
than N GPUs it says so on stderr and runs on the GPUs that exist (`n_gpus` reports the real count).

Prints ONE JSON line on rank 0.  `value` = clips/s over all N GPUs with clips already resident in
HBM.  `roofline` is for the dominant kernel class (all vtx_gemm_nt launches): achieved =
algorithmic FLOPs (2*M*N*K per launch) / launch time measured with HIP events on the launch stream
inside the timed region, against the 2.5 PFLOP/s dense bf16 MFMA peak.  After the timed region a
short instrumented pass (not part of `value`) times every kernel class the same way:
`gemm_shapes` (each GEMM shape of the step with its own algorithmic bytes, bound and fraction of
that bound) and `roofline_hbm` (LayerNorm, attention cores, patch gather, HOG against 8 TB/s).
`cpu_baseline` times the CPU oracle (oracle/, fp32, the host cores this process may use) on a
bounded sample of the same workload -- a reported baseline only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOPS_FWD_BWD_PER_CLIP = {8: 1.175e12, 16: 2.352e12, 2: 0.2937e12}
PEAK_BF16 = 2500.0     # TFLOP/s dense (MI355X_MICROARCH.md)
PEAK_HBM = 8.0         # TB/s (spec; ~6.3 TB/s is what a streaming copy reaches)
PMC_ROUNDS = ('round6_', 'round5_', 'round4_', 'round3_', 'round2_')     # committed rocprofv3 counter passes of the default command, newest first


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=96, help='clips per GPU (weak scaling)')
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--stream', default='bf16', choices=['bf16', 'fp32', 'fp32+grad'],
                    help="residual stream of the bf16 path: bf16 (default, the headline), fp32 = vtx.set_stream('fp32'), the exact stream, or fp32+grad = its gradient in float32 too (DESIGN.md 3)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true', help='skip the instrumented per-kernel-class pass')
    ap.add_argument('--no-other-configs', action='store_true',
                    help="skip the 5-step mini-runs of north_star's other shapes after the timed region (other_configs in the JSON line)")
    ap.add_argument('--no-optimizer', action='store_true')
    ap.add_argument('--no-direct-grads', action='store_true',
                    help='parameter gradients through autograd accumulation instead of straight into the buckets')
    ap.add_argument('--optimizer', default='fused', choices=['fused', 'torch'],
                    help='fused = vtx multi-tensor SGD-nesterov kernel; torch = torch.optim.SGD (foreach)')
    return ap.parse_args()


def _host_cores():
    """CPUs this process may really use: min(affinity, cgroup CPU quota)."""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_subprocess(frames, budget_s=150):
    """Run cpu_baseline() in a child process with a hard wall-clock bound so that the default
    bench run always finishes within minutes, whatever the GPU box's host looks like."""
    code = ('import sys, json; sys.path.insert(0, %r); import bench; '
            'print("CPU_BASELINE " + json.dumps(bench.cpu_baseline(%d)))' % (ROOT, frames))
    try:
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=budget_s)
        for line in r.stdout.splitlines():
            if line.startswith('CPU_BASELINE '):
                return json.loads(line[len('CPU_BASELINE '):])
        why = 'no result: ' + (r.stderr.strip().splitlines() or ['?'])[-1][:200]
    except subprocess.TimeoutExpired:
        why = 'exceeded the %d s budget' % budget_s
    return {'value': None, 'unit': 'clips/s', 'cores': min(_host_cores(), 32), 'kind': 'port',
            'sample': 'oracle/vt_oracle.py TimeSformer-B fwd+bwd, batch 1: ' + why}


def cpu_baseline(frames, steps=3, budget_s=60.0):
    """CPU oracle (fp32, torch CPU kernels on the host cores this process may use, at most 32),
    train-mode fwd+bwd of the same model on one clip: 1 warm-up + `steps` timed (SURVEY.md 8(d) asks
    for >= 3; fewer only if they would exceed `budget_s` seconds, and the count is reported)."""
    from oracle import synth, vt_oracle as O
    import video_transformer as V
    torch.set_num_threads(min(_host_cores(), 32))
    sd = synth.synth_state_dict(synth.shapes_of(V.TimeSformer(num_frames=frames)), 0)
    for v in sd.values():
        v.requires_grad_(True)
    x = synth.synth_clip(1, frames, seed=1)
    times = []
    t_begin = time.perf_counter()
    for i in range(steps + 1):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        torch.manual_seed(i)
        y = O.timesformer_forward(sd, x, frames, training=True)
        y.sum().backward()
        times.append(time.perf_counter() - t0)
        if len(times) >= 2 and time.perf_counter() - t_begin + times[-1] > budget_s:
            break
    timed = times[1:]
    t = sum(timed) / len(timed)
    return {'value': round(1.0 / t, 4), 'unit': 'clips/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'oracle/vt_oracle.py TimeSformer-B {frames}x224^2 fp32 train fwd+bwd, batch 1, '
                      f'{len(timed)} timed steps after 1 warm-up ({sum(timed):.1f} s of CPU work)'}


def pmc_traffic_per_launch(B, args):
    """HBM-side bytes per vtx_gemm_nt launch REPLAYED from the committed rocprofv3 PMC passes of THIS command
    (profiles/round<N>_pmc_{FETCH,WRITE}_SIZE_b<B>.txt: separate --pmc passes, KB per dispatch summed over
    the listed dispatches).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced
    streams on gfx950; WRITE_SIZE is uncalibrated and taken as is.  None when no committed pass matches
    the configuration (the counters cannot be collected from inside the timed run)."""
    if args.precision != 'bf16' or args.frames != 8:
        return None
    here = os.path.dirname(os.path.abspath(__file__))
    tot = 0.0
    for name, mult in ((f'pmc_FETCH_SIZE_b{B}.txt', 2.0), (f'pmc_WRITE_SIZE_b{B}.txt', 1.0)):
        try:
            kb, rows = 0.0, 0.0
            path = next(pth for pth in (os.path.join(here, 'profiles', r + name) for r in PMC_ROUNDS) if os.path.isfile(pth))
            for line in open(path):
                if 'nt_bf16_pp_kernel' in line:            # one line per template instantiation (names are cut on the left): pool them
                    f = dict(kv.split('=') for kv in line.split() if '=' in kv)
                    kb += float(f['total'])
                    rows += float(f['rows'])
            if rows <= 0:
                return None
            tot += mult * kb * 1024.0 / rows
        except (OSError, KeyError, ValueError, StopIteration):
            return None
    return round(tot, 0) if tot > 0 else None


def pmc_mfma_busy(B, args):
    """Matrix-pipe occupancy per GEMM class REPLAYED from the committed `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
    GRBM_GUI_ACTIVE` pass of THIS command (profiles/round<N>_pmc_MFMA_BUSY_b<B>.txt, kernel names pooled over template
    instantiations).  SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of every SIMD's matrix pipe (32 per v_mfma_f32_32x32x16_bf16:
    MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs.  busy = MFMA_BUSY / (GRBM / 8 * 1024 SIMDs): the
    fraction of SIMD-cycles AT THE CLOCK THE KERNEL RAN AT with the matrix pipe busy -- roofline.frac prices the same work
    against the nominal 2.4 GHz peak, the ratio of the two is the clock the power budget allowed."""
    if args.precision != 'bf16' or args.frames != 8:
        return None
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        path = next(pth for pth in (os.path.join(here, 'profiles', r + f'pmc_MFMA_BUSY_b{B}.txt') for r in PMC_ROUNDS) if os.path.isfile(pth))
    except StopIteration:
        return None
    acc = {}
    for line in open(path):
        cls = 'gemm_nt' if 'nt_bf16_pp_kernel' in line else 'gemm_tn' if 'tn_bf16_pp_kernel' in line else None
        if cls is None:
            continue
        try:
            f = dict(kv.split('=') for kv in line.split() if '=' in kv)
            counter = next(c for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE') if c in line)
            acc.setdefault(cls, {}).setdefault(counter, 0.0)
            acc[cls][counter] += float(f['total'])
        except (StopIteration, KeyError, ValueError):
            continue
    out = {c: round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0), 4)
           for c, v in acc.items() if v.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in v}
    if not out:
        return None
    out['source'] = 'replayed: ' + os.path.relpath(path, here)
    return out


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks of this script on 127.0.0.1."""
    have = torch.cuda.device_count()
    n = args.gpus
    if have < n:
        sys.stderr.write(f'bench.py: --gpus {n} requested but this node has {have} GPU(s); running on {max(have, 1)}\n')
        n = max(have, 1)
    if n <= 1:
        return False
    argv = [a for a in sys.argv[1:]]
    for i, a in enumerate(argv):                      # the ranks get the real count
        if a == '--gpus':
            argv[i + 1] = str(n)
        elif a.startswith('--gpus='):
            argv[i] = f'--gpus={n}'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    rc = subprocess.run(cmd, env=env).returncode
    sys.exit(rc)


def gemm_shape_table(per_shape, peak_tf):
    """[{shape, launches_per_step, avg_us, tflops, alg_mb, tb_per_s, bound, frac}] from ops.profile_stop() entries.  Launches that
    differ only in M by less than 15 % are pooled into one row `Mmin..MmaxxNxK` (the FFN runs on the clips DropPath keeps: a
    different M per layer and step; temporal / spatial row counts 150528 / 150624 / 151296)."""
    groups = {}
    for key, (n, ms, fl, by) in per_shape.items():
        dims, _, epi = key.partition('+')
        try:
            M, N, K = (int(v) for v in dims.split('x'))
        except ValueError:
            groups[(key, None, None, 0)] = [key, key, n, ms, fl, by]
            continue
        for g in groups:
            if g[1:3] == (N, K) and g[0] == epi and isinstance(g[3], int) and g[3] and abs(M - g[3]) <= 0.15 * g[3]:
                v = groups[g]
                v[0], v[1] = min(v[0], M), max(v[1], M)
                v[2] += n; v[3] += ms; v[4] += fl; v[5] += by
                break
        else:
            groups[(epi, N, K, M)] = [M, M, n, ms, fl, by]
    rows = []
    for (epi, N, K, _), (m0, m1, n, ms, fl, by) in sorted(groups.items(), key=lambda kv: -kv[1][3]):
        if N is None:
            shape = m0
        else:
            shape = (f'{m0}x{N}x{K}' if m0 == m1 else f'{m0}..{m1}x{N}x{K}') + (f'+{epi}' if epi else '')
        t = ms * 1e-3 / n
        t_m, t_h = fl / n / (peak_tf * 1e12), by / n / (PEAK_HBM * 1e12)
        rows.append({'shape': shape, 'launches': n, 'avg_us': round(t * 1e6, 1), 'tflops': round(fl / n / t / 1e12, 1),
                     'alg_mb': round(by / n / 1e6, 1), 'tb_per_s': round(by / n / t / 1e12, 2),
                     'bound': 'mfma' if t_m >= t_h else 'hbm', 'frac': round(max(t_m, t_h) / t, 3)})
    return rows


def other_configs(dev, model8, head8, budget_s=40.0):
    """north_star's other shapes on the driver-witnessed line: 5-step mini-runs AFTER the timed region (never part of `value`),
    same step structure as the headline (bf16, buckets with direct gradients, fused optimizer, synthetic clips resident in HBM),
    under a hard wall-clock budget -- a configuration that would start beyond it is reported as skipped."""
    import vtx
    from vtx import dp, optim, functions as F_
    import transformer as T
    import video_transformer as V
    t_begin = time.perf_counter()
    out = []

    def run(workload, build, batch, frames, flops_per_clip, extra=None, steps=5, warmup=2):
        if time.perf_counter() - t_begin > budget_s:
            out.append({'workload': workload, 'skipped': 'time budget of %.0f s for other_configs used up' % budget_s})
            return
        vtx.set_precision('bf16')
        torch.manual_seed(0)
        stepfn, cleanup = build(batch, frames)
        try:
            for _ in range(warmup):
                stepfn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                stepfn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            row = {'workload': workload, 'clips_per_gpu': batch, 'clips_per_s': round(batch / dt, 2), 'ms_per_step': round(dt * 1e3, 3),
                   'steps': steps, 'mfma_frac_nominal': round(batch / dt * flops_per_clip / 1e12 / PEAK_BF16, 4) if flops_per_clip else None}
            row.update(extra or {})
            out.append(row)
        finally:
            cleanup()
            torch.cuda.empty_cache()

    def classifier(model, head):
        def build(batch, frames):
            params = list(model.parameters()) + list(head.parameters())
            buckets = dp.GradBuckets(params, direct=True)
            opt = optim.FusedSGD(buckets, lr=1e-4, momentum=0.9, nesterov=True)
            x = torch.randn(batch, frames, 3, 224, 224, device=dev)
            y = torch.randint(0, 400, (batch,), device=dev)

            def stepfn():
                buckets.zero()
                F_.SoftmaxXentFn.apply(head(model(x)), y).backward()
                buckets.finish()
                opt.step()
            return stepfn, buckets.remove
        return build

    def fresh(ctor):
        m = ctor().to(dev).train()
        return m, T.ClassificationHead(400, m.embed_dims).to(dev).train()

    # the headline model at the batch the reference trains at (8 clips per GPU, demo log) and at 32 clips
    for b in (8, 32):
        run('TimeSformer-B divided_space_time, 8x3x224x224, bf16, fwd+CE+bwd+SGD (headline model, %d clips per GPU)' % b,
            classifier(model8, head8), b, 8, FLOPS_FWD_BWD_PER_CLIP[8], steps=12, warmup=3)
    # the headline model and batch with the EXACT residual stream (vtx.set_stream('fp32'): the running sum of the residual stream in
    # float32 as under the reference's autocast) -- what the accuracy mode costs
    # ('fp32+grad': the stream's gradient in float32 through the backward as well -- vtx_layernorm_bwd_g32)
    try:
        for mode, what in (('fp32', 'the exact float32 residual stream'), ('fp32+grad', 'the exact float32 residual stream and its float32 gradient')):
            vtx.set_stream(mode)
            run(f"TimeSformer-B divided_space_time, 8x3x224x224, bf16 with {what} (vtx.set_stream('{mode}')), fwd+CE+bwd+SGD, 96 clips per GPU",
                classifier(model8, head8), 96, 8, FLOPS_FWD_BWD_PER_CLIP[8], steps=6, warmup=2)
    finally:
        vtx.set_stream('bf16')
    run('TimeSformer-B divided_space_time, 16x3x224x224, bf16, fwd+CE+bwd+SGD (north_star second shape)',
        classifier(*fresh(lambda: V.TimeSformer(num_frames=16))), 48, 16, FLOPS_FWD_BWD_PER_CLIP[16])
    run('ViViT-B fact_encoder, Conv3d tubelet 2, 16x3x224x224, bf16, fwd+CE+bwd+SGD (BASELINE configs[2])',
        classifier(*fresh(lambda: V.ViViT(num_frames=16))), 32, 16, 0.850e12, steps=8)

    # BASELINE configs[4]: TimeSformer-L (D 1024, 16 heads, 24 layers) on 96-frame clips -- 18 817 tokens per clip, every activation
    # stored (no recompute: 2.1 GB per layer and clip, ~84 GB at 4 clips of the 288).  FLOPs per clip fwd+bwd: the Linears
    # (17 D^2 per token and layer) + the attention cores (tools/other_configs.py::timesformer_l96).
    tokens_l, D_l, layers_l = 196 * 96 + 1, 1024, 24
    flops_l = 6.0 * tokens_l * (17 * D_l * D_l) * layers_l + 3.0 * layers_l * (4.0 * tokens_l * 96 * D_l + 4.0 * tokens_l * 197 * D_l)
    run('TimeSformer-L divided_space_time (D 1024, 16 heads, 24 layers), 96x3x224x224, bf16, fwd+CE+bwd+SGD, all activations stored '
        '(BASELINE configs[4], one GPU)',
        classifier(*fresh(lambda: V.TimeSformer(num_frames=96, embed_dims=1024, num_heads=16, num_transformer_layers=24))),
        4, 96, flops_l, steps=3, warmup=1)

    def maskfeat(batch, frames):
        from vtx import ops
        m = V.MaskFeat(pool_q_stride_size=[[1, 1, 2, 2], [3, 1, 2, 2]], feature_dim=216).to(dev).train()
        opt = optim.FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.05, clip_grad=0.02)
        x = torch.randn(batch, 16, 3, 224, 224, device=dev)
        fr = torch.randint(0, 256, (batch * 3, 224, 224, 3), dtype=torch.uint8, device=dev)
        mask = torch.zeros(batch, 8, 14, 14, dtype=torch.int32)
        mask[:, 2:4, 3:9, 2:10] = 1
        mask[:, 6, 5:12, 5:12] = 1
        mask = mask.to(dev)
        markers = [[[2, 2], [6, 1]]] * batch

        def stepfn():
            hog = ops.hog_fwd(fr)                       # HOG targets on the device, inside the step
            target = torch.zeros(batch, 16, 14, 14, 108, dtype=torch.float64, device=dev)
            target[:, 6] = hog[0::3]
            target[:, 13] = hog[1::3]
            m.zero_grad(set_to_none=True)
            _, loss = m(x, target, mask, markers)
            loss.backward()
            opt.step()
        return stepfn, (lambda: None)
    run('MaskFeat: MViT-B + HOG-target masked MSE, 16x3x224x224, bf16, fwd+bwd+AdamW, HOG targets computed in the step (BASELINE configs[3], one GPU)',
        maskfeat, 32, 16, 3 * 70.6e9, extra={'parity': 'unpinned (the MViT oracle restates pytorchvideo 0.1.3, which is on no disk: DESIGN.md 5)'})
    return out


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        respawn_under_torchrun(args)                  # does not return when it spawned the ranks
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus and rank == 0:
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}\n')
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    assert local < torch.cuda.device_count(), f'rank {rank}: no GPU {local} on this node'
    # host threads: never more than the cgroup CPU quota allows (an oversubscribed ATen pool gets the
    # whole process CFS-throttled, which shows up as ~90 ms launch stalls)
    torch.set_num_threads(max(1, min(_host_cores() // max(world, 1), 16)))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    force_dp = os.environ.get('VTX_FORCE_DP', '0') == '1'     # 1-rank RCCL group: exercises the DP path on one GPU
    import __graft_entry__ as ge
    ge.ensure_built()
    import vtx
    if world > 1 or force_dp:
        from vtx import dp as _dp
        _dp.init_process_group(dev, rank, world)   # launcher's MASTER_*; a one-rank group picks (and retries) its own port
    from vtx import dp, ops, optim, functions as F_
    import transformer as T
    import video_transformer as V

    vtx.set_precision(args.precision)
    vtx.set_stream(args.stream)
    torch.manual_seed(0)
    model = V.TimeSformer(num_frames=args.frames)
    head = T.ClassificationHead(400, model.embed_dims)
    with torch.no_grad():                               # temporal_fc is zero-initialised: give it weight
        for blk in model.transformer_layers.layers:
            blk.attentions[0].temporal_fc.weight.normal_(0, 0.02)
    model.to(dev).train()
    head.to(dev).train()
    params = list(model.parameters()) + list(head.parameters())
    dp.broadcast_parameters(model)
    dp.broadcast_parameters(head)
    # gradients live in flat per-layer buckets: the all-reduce units for N > 1 and the multi-tensor
    # optimizer's operands for any N
    buckets = dp.GradBuckets(params, force_comm=force_dp, direct=not args.no_direct_grads)
    opt = None
    if not args.no_optimizer:
        if args.optimizer == 'fused':
            opt = optim.FusedSGD(buckets, lr=1e-4, momentum=0.9, nesterov=True)
        else:
            opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9, nesterov=True)

    B = args.batch
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    x = torch.randn(B, args.frames, 3, 224, 224, generator=g).to(dev)
    labels = torch.randint(0, 400, (B,), generator=g).to(dev)

    def step():
        buckets.zero()
        logits = head(model(x))
        loss = F_.SoftmaxXentFn.apply(logits, labels)        # vtx_softmax_xent_fwd / _bwd (csrc/head.hip)
        loss.backward()
        buckets.finish()
        if opt is not None:
            opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.profile_start(('gemm_nt',))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ops.profile_stop()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.detach())
    rccl_world = dist.get_world_size() if dist.is_initialized() else 1

    # ---- instrumented pass (outside the timed region): every kernel class, HIP events per launch ----
    classes = None
    if not args.no_breakdown and rank == 0:
        names = ('gemm_nt', 'gemm_tn', 'wprod', 'attn_fwd_time', 'attn_bwd_time', 'attn_fwd_space', 'attn_bwd_space', 'ln_fwd', 'ln_bwd',
                 'colsum', 'patch_rows', 'hog')
        nb = 2
        frames_u8 = torch.randint(0, 256, (1024, 224, 224, 3), dtype=torch.uint8, device=dev)   # MaskFeat HOG targets of 64 clips
        ops.hog_fwd(frames_u8)                          # builds / uploads the magnitude table once
        ops.profile_start(names)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nb):
            step()
        torch.cuda.synchronize()
        step_ms = (time.perf_counter() - t1) / nb * 1e3
        for _ in range(nb):
            ops.hog_fwd(frames_u8)
        torch.cuda.synchronize()
        classes = ops.profile_stop()
        classes['_step_ms'] = step_ms
        classes['_steps'] = nb
    others = None
    if world == 1 and not force_dp and not args.no_other_configs and args.precision == 'bf16' and args.frames == 8 and args.stream == 'bf16':
        buckets.remove()                                # the mini-runs build their own buckets over the same parameters
        del x
        torch.cuda.empty_cache()
        try:
            others = other_configs(dev, model, head)
        except Exception as e:                          # the headline line must survive anything that happens here
            others = [{'error': repr(e)[:300]}]
    if world > 1:
        dist.barrier()

    if rank == 0:
        peak = PEAK_BF16 if args.precision == 'bf16' else 157.3
        clips = world * B * args.steps
        value = clips / elapsed
        n, ms, flops, nbytes = ops.profile_totals(prof)['gemm_nt']
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        out = {
            'metric': 'clips/sec TimeSformer-B divided_space_time %dx3x224x224 fwd+bwd (whole job)' % args.frames,
            'value': round(value, 3), 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': 'TimeSformer-B divided_space_time, %d frames x 3x224x224, %s, fwd+CE+bwd%s, '
                                   'random-init weights' % (args.frames, args.precision,
                                                            '' if args.no_optimizer else '+SGD(nesterov, %s)' % args.optimizer),
                       'clips_per_gpu': B, 'global_batch': B * world, 'parallelism': 'dp%d' % world, 'residual_stream': args.stream,
                       'grad_exchange': 'RCCL all-reduce (world size %d), per-layer fp32 buckets overlapped with backward' % rccl_world
                       if (world > 1 or force_dp) else 'none'},
            'clips_per_sec_per_gpu': round(value / world, 3),
            'model_tflops_per_gpu': round(value / world * FLOPS_FWD_BWD_PER_CLIP.get(args.frames, 0) / 1e12, 2),
            'mfma_frac_whole_step_nominal': round(value / world * FLOPS_FWD_BWD_PER_CLIP.get(args.frames, 0) / 1e12 / PEAK_BF16, 4),
            'model_flops_note': 'model_tflops / mfma_frac_whole_step_nominal price the NOMINAL FLOPs of the reference graph (BASELINE.md '
                                'section 3); mfma_frac_whole_step prices the FLOPs the step EXECUTES (matrix products of every '
                                'vtx_gemm_nt / vtx_gemm_tn / vtx_wprod / attention launch of one instrumented step: the merged attn.proj + '
                                'temporal_fc GEMM (DESIGN.md 4.4) and the clips DropPath drops at the FFN (4.5) are not computed) over the '
                                'timed ms_per_step. roofline.* counts executed FLOPs only.',
            'final_loss': round(final_loss, 4),
            'roofline': {'kernel': 'gemm_nt_bf16_pp_kernel (all vtx_gemm_nt launches)' if args.precision == 'bf16' else 'gemm_nt_f32_kernel',
                         'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(achieved / peak, 4),
                         'traffic': pmc_traffic_per_launch(B, args),
                         'traffic_source': 'replayed: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command under profiles/ '
                                           '(2 x FETCH_SIZE + WRITE_SIZE per launch), not counted in this run',
                         'mfma_busy': pmc_mfma_busy(B, args),
                         'launches': n, 'avg_launch_us': round(ms / max(n, 1) * 1e3, 2),
                         'flops_per_launch_avg': round(flops / max(n, 1), 0),
                         'algorithmic_bytes_per_launch_avg': round(nbytes / max(n, 1), 0)},
        }
        if classes is not None:
            nb = classes.pop('_steps')
            out['instrumented_step_ms'] = round(classes.pop('_step_ms'), 3)
            tot = ops.profile_totals(classes)
            out['gemm_shapes'] = {'nt': gemm_shape_table(classes['gemm_nt'], peak), 'tn': gemm_shape_table(classes['gemm_tn'], peak)}
            for k in ('nt', 'tn'):
                for r in out['gemm_shapes'][k]:
                    r['launches_per_step'] = r.pop('launches') // nb
            mm = ('gemm_nt', 'gemm_tn', 'wprod', 'attn_fwd_time', 'attn_bwd_time', 'attn_fwd_space', 'attn_bwd_space')
            executed = sum(tot[c][2] for c in mm if c in tot) / nb
            out['executed_matrix_tflop_per_step'] = round(executed / 1e12, 3)
            out['mfma_frac_whole_step'] = round(executed / (elapsed / args.steps) / 1e12 / peak, 4)
            out['gemm_tn_roofline'] = {'achieved': round(tot['gemm_tn'][2] / (tot['gemm_tn'][1] * 1e-3) / 1e12, 2), 'peak': peak,
                                       'unit': 'TFLOP/s', 'frac': round(tot['gemm_tn'][2] / (tot['gemm_tn'][1] * 1e-3) / 1e12 / peak, 4),
                                       'ms_per_step': round(tot['gemm_tn'][1] / nb, 3)}
            hbm = []
            label = {'ln_fwd': 'ln_fwd_kernel', 'ln_bwd': 'ln_bwd_kernel', 'attn_fwd_time': 'attn_fwd_small_kernel (temporal, T tokens)',
                     'attn_bwd_time': 'attn_bwd_small_kernel (temporal)', 'attn_fwd_space': 'attn_fwd_*_mfma_kernel (spatial, 197 tokens)',
                     'attn_bwd_space': 'attn_bwd_*_mfma_kernel (spatial)', 'patch_rows': 'patch_rows_kernel (clip gather)',
                     'colsum': 'colsum_kernel', 'hog': 'hog_kernel (1024 frames 224x224x3 uint8 -> float64 features)'}
            for c, (cn, cms, cfl, cby) in tot.items():
                if c in label and cn:
                    a = cby / (cms * 1e-3) / 1e12
                    per_step = None if c == 'hog' else round(cms / nb, 3)
                    hbm.append({'kernel': label[c], 'bound': 'hbm', 'achieved': round(a, 3), 'peak': PEAK_HBM, 'unit': 'TB/s',
                                'frac': round(a / PEAK_HBM, 4), 'avg_launch_us': round(cms / cn * 1e3, 1),
                                'algorithmic_mb_per_launch': round(cby / cn / 1e6, 2), 'ms_per_step': per_step})
            out['roofline_hbm'] = hbm
        if others is not None:
            out['other_configs'] = others
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_subprocess(args.frames)
    else:
        out = None
    # tear the process group down first and flush C stdio (RCCL prints its version banner through it), so
    # that the JSON line is the LAST thing this job writes to stdout
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
