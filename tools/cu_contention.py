"""What a LIVE collective costs the step, measured on one GPU.

An RCCL all-reduce occupies a few compute units for as long as it is in flight.  The data-parallel path of bench.py overlaps its
bucket all-reduces with backward, so the step's kernels then run on fewer CUs.  On a one-GPU box no collective has a peer, so this
tool stands in for one: `cu_hold_kernel` (tools/micro/cu_hold.hip) parks one 256-thread workgroup on each of H compute units from a
SIDE stream -- a wave on every SIMD of the CU, so the 512-thread / 256-register GEMM workgroups cannot be placed there, exactly
what a channel of a collective does to them -- for the whole measurement, and the bench step (same model, buckets, optimizer as
bench.py) is timed with HIP events per GEMM launch.  H = 0 / 8 / 16 / 32.  The hold is CONTINUOUS here; in the real job a
collective is in flight for 3 - 9 % of the backward, so the step-level cost scales by that duty cycle.

    python tools/cu_contention.py [--batch 96] [--steps 6] [--holds 0,8,16,32,0] [--tn-cus 256,240]

--tn-cus: values of the option `tn_cus` (CUs the weight-gradient kernel's slab split is sized for; 240 = room for 16 held CUs) to
repeat the sweep with.
"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'videotransformer-pytorch_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402


def load_holder():
    so = os.path.join(ROOT, 'tools', 'micro', 'libcuhold.so')
    src = os.path.join(ROOT, 'tools', 'micro', 'cu_hold.hip')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', '-o', so, src], check=True)
    lib = ctypes.CDLL(so)
    lib.cu_hold_launch.restype = ctypes.c_int
    lib.cu_hold_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def main():
    opt = lambda name, d: sys.argv[sys.argv.index(name) + 1] if name in sys.argv else d      # noqa: E731
    B, steps = int(opt('--batch', 96)), int(opt('--steps', 6))
    holds = [int(v) for v in opt('--holds', '0,8,16,32,0').split(',')]
    tn_cus_list = [int(v) for v in opt('--tn-cus', '256').split(',')]
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    import vtx
    from vtx import dp, ops, optim, functions as F_
    import transformer as T
    import video_transformer as V
    vtx.set_precision('bf16')
    torch.manual_seed(0)
    model = V.TimeSformer(num_frames=8)
    head = T.ClassificationHead(400, model.embed_dims)
    with torch.no_grad():
        for blk in model.transformer_layers.layers:
            blk.attentions[0].temporal_fc.weight.normal_(0, 0.02)
    model.to(dev).train()
    head.to(dev).train()
    params = list(model.parameters()) + list(head.parameters())
    buckets = dp.GradBuckets(params, direct=True)
    opt_ = optim.FusedSGD(buckets, lr=1e-4, momentum=0.9, nesterov=True)
    g = torch.Generator(device='cpu').manual_seed(1234)
    x = torch.randn(B, 8, 3, 224, 224, generator=g).to(dev)
    labels = torch.randint(0, 400, (B,), generator=g).to(dev)

    def step():
        buckets.zero()
        F_.SoftmaxXentFn.apply(head(model(x)), labels).backward()
        buckets.finish()
        opt_.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    lib = load_holder()
    side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
    print(f'TimeSformer-B 8x224^2, {B} clips, bf16 step (fwd + CE + bwd + SGD), {steps} timed steps per setting; holder = one 256-thread '
          f'workgroup per held CU on a side stream for the whole measurement')
    print(f'{"held CUs":>8s} {"distinct CUs":>12s} {"ms/step":>8s} {"clips/s":>8s} {"vs 0":>7s} | {"NT us":>7s} {"vs 0":>6s} | {"TN us":>7s} {"vs 0":>6s} | TN per shape (us)')
    for tn_cus in tn_cus_list:
        vtx.set_option('tn_cus', str(tn_cus))
        print(f'-- option tn_cus = {tn_cus}')
        base = None
        for H in holds:
            stop = torch.zeros(1, dtype=torch.int32, device=dev)
            census = torch.zeros(2 * max(H, 1), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            if H:
                rc = lib.cu_hold_launch(H, 20.0, stop.data_ptr(), census.data_ptr(), side.cuda_stream)
                assert rc == 0, rc
                time.sleep(0.05)                                  # the holders are resident before the first kernel of the step
            step()                                               # one untimed step under the same condition
            ops.profile_start(('gemm_nt', 'gemm_tn'))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                step()
            e1.record()
            e1.synchronize()
            prof = ops.profile_stop()
            with torch.cuda.stream(side2):
                stop.fill_(1)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            tot = ops.profile_totals(prof)
            nt_us = tot['gemm_nt'][1] / tot['gemm_nt'][0] * 1e3
            tn_us = tot['gemm_tn'][1] / tot['gemm_tn'][0] * 1e3
            cz = census.cpu().numpy().astype('uint32')
            distinct = len({(int(cz[2 * i]) & 0xF, int(cz[2 * i + 1]) & 0xFF00) for i in range(H)}) if H else 0
            shapes = {}
            for key, (n, tms, fl, by) in prof['gemm_tn'].items():
                _, n1, n2 = key.split('x')
                r = shapes.setdefault(f'{n1}x{n2}', [0, 0.0])
                r[0] += n
                r[1] += tms
            per = '  '.join(f'{k} {v[1] / v[0] * 1e3:.0f}' for k, v in sorted(shapes.items()))
            if base is None:
                base = (ms, nt_us, tn_us)
            print(f'{H:8d} {distinct:12d} {ms:8.2f} {B / ms * 1e3:8.1f} {ms / base[0]:7.3f} | {nt_us:7.1f} {nt_us / base[1]:6.3f} | {tn_us:7.1f} {tn_us / base[2]:6.3f} | {per}',
                  flush=True)


if __name__ == '__main__':
    main()
