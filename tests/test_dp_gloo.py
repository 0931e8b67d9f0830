"""N>1 path on CPU: world_size-2 gloo run of the gradient-bucket exchange (vtx/dp.py)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, 'videotransformer-pytorch_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vtx import dp
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 8),
                                torch.nn.LayerNorm(8))
    dp.broadcast_parameters(model)
    ref = [p.detach().clone() for p in model.parameters()]
    buckets = dp.GradBuckets(model.parameters(), bucket_bytes=1024)     # forces several buckets
    assert len(buckets.buckets) >= 2
    g = torch.Generator().manual_seed(7)
    data = torch.randn(8, 16, generator=g)                              # global batch of 8 "clips"
    mine = dp.shard_clips(8, rank, world)
    for step in range(2):
        buckets.zero()
        loss = model(data[mine]).pow(2).sum() / 8
        loss.backward()
        buckets.finish()
    got = [p.grad.clone() for p in model.parameters()]
    # single-process reference on the full batch with rank 0's (broadcast) weights
    m2 = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 8), torch.nn.LayerNorm(8))
    for p, r in zip(m2.parameters(), ref):
        p.data.copy_(r)
    (m2(data).pow(2).sum() / 8 / world).backward()                      # mean over ranks of per-rank sums
    err = max((a - b.grad).abs().max().item() for a, b in zip(got, m2.parameters()))
    same_w = all(torch.equal(a, b) for a, b in zip(ref, [p.detach() for p in model.parameters()]))
    ret[rank] = (err, same_w, len(buckets.buckets))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    world = 2
    import socket
    sock = socket.socket()                              # a free port picked by the kernel (a fixed 295xx one collides with a
    sock.bind(('127.0.0.1', 0))                         # leftover store of an earlier run: EADDRINUSE)
    port = sock.getsockname()[1]
    sock.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        err, same_w, nb = ret[r]
        assert err < 1e-6, f'rank {r}: averaged gradient mismatch {err}'
        assert same_w


def _worker_unused(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, 'videotransformer-pytorch_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vtx import dp
    torch.manual_seed(5)
    a, b, c = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    buckets = dp.GradBuckets(params, bucket_bytes=64)
    buckets.zero()
    x = torch.ones(2, 4)
    y = a(x)
    if rank == 0:                                       # a data-dependent branch: only rank 0 runs b; nobody runs c
        y = y + b(x)
    y.sum().backward()
    buckets.finish()
    local_unfired = {id(p) for bk in buckets.buckets for p in bk['params'] if id(p) not in buckets._fired}
    glob = {id(p) for p in buckets.unfired()}
    ret[rank] = (glob == {id(p) for p in c.parameters()},
                 local_unfired == ({id(p) for p in c.parameters()} | ({id(p) for p in b.parameters()} if rank else set())),
                 b.weight.grad.abs().sum().item())
    dist.destroy_process_group()


def test_unfired_parameters_are_agreed_across_ranks():
    """A parameter used on ONE rank only is not "unused": every rank must update it with the averaged gradient (DDP's
    find_unused_parameters all-reduces its used bitmap); only parameters no rank used are skipped."""
    import socket
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_unused, args=(2, port, ret), nprocs=2, join=True)
    for r in range(2):
        agreed, local_ok, gsum = ret[r]
        assert agreed, f'rank {r}: unfired() must name exactly the parameters unused on every rank'
        assert local_ok
        assert abs(gsum - 16 * 2 * 0.5) < 1e-6, f'rank {r}: mean gradient of the half-used weight {gsum}'
    assert ret[0][2] == ret[1][2]


def _worker_static_unused(rank, world, port, ret):
    import contextlib
    import io
    sys.path.insert(0, os.path.join(ROOT, 'videotransformer-pytorch_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vtx import dp
    torch.manual_seed(5)
    a, b, c = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())      # c: registered last = buckets 0 and 1
    buckets = dp.GradBuckets(params, bucket_bytes=64, static_unused=True)
    nb = len(buckets.buckets)
    x = torch.ones(2, 4) * (rank + 1)
    out = {}
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        for step in range(3):
            buckets.zero()
            (a(x) + b(x)).sum().backward()                   # nobody ever runs c
            out[f'next{step}'] = buckets._next              # buckets that went out DURING backward
            buckets.finish()
            buckets.unfired()
    out['warnings'] = err.getvalue().count('finish() had to issue')
    out['nb'] = nb
    out['grad'] = a.weight.grad.clone()
    # the graph changes after all: c gets a gradient although every rank had agreed it is unused.  Its buckets (0 and 1, all of
    # their parameters known unused) go out as soon as the first gradient of the step arrives, so c's gradient comes too late
    buckets.zero()
    try:
        (c(a(x)) + b(x)).sum().backward()                   # b's gradients arrive first (created last), then c's
        out['late'] = 'no error'
    except RuntimeError as e:
        out['late'] = 'raised' if 'static_unused' in str(e) else repr(e)
    ret[rank] = out
    if out['late'] != 'raised':
        buckets.finish()
    dist.destroy_process_group()


def test_static_unused_parameters_release_their_buckets():
    """buckets go out strictly in order, so ONE parameter without a gradient (here: a whole unused layer registered
    last = buckets 0 and 1) holds every bucket back until finish() -- no overlap with backward, reported once on stderr.  With
    static_unused=True the set unfired() agrees on across ranks no longer counts from the next step on: every bucket goes out
    during backward again.  A parameter of that set that does get a gradient later raises instead of being lost."""
    import socket
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_static_unused, args=(2, port, ret), nprocs=2, join=True)
    for r in range(2):
        o = ret[r]
        assert o['next0'] == 0, 'step 0: the unused layer holds every bucket back'
        assert o['warnings'] == 1, 'the held buckets are reported exactly once'
        assert o['next1'] == o['nb'] and o['next2'] == o['nb'], (o['next1'], o['next2'], o['nb'])
        assert o['late'] == 'raised', o['late']
    # mean over ranks of d/dW sum(a(x)) = mean of x summed over the batch: (2 * 1 + 2 * 2) / 2 = 3 per weight column
    assert torch.allclose(ret[0]['grad'], torch.full((4, 4), 3.0)) and torch.equal(ret[0]['grad'], ret[1]['grad'])


def _worker_bf16_wire(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, 'videotransformer-pytorch_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vtx import dp
    torch.manual_seed(3)
    lin = torch.nn.Linear(8, 8)
    buckets = dp.GradBuckets(lin.parameters(), comm_dtype=torch.bfloat16)
    ptrs = []
    try:
        for step in range(2):
            buckets.zero()
            (lin(torch.ones(2, 8) * (rank + 1 + step))).sum().backward()
            buckets.finish()
            ptrs.append(buckets.buckets[0]['comm_buf'].data_ptr())
        ret[rank] = ('ok', lin.weight.grad.clone(), ptrs)
    except RuntimeError as e:                           # a gloo build without bfloat16 reductions
        ret[rank] = ('unsupported: ' + str(e)[:80], None, ptrs)
    dist.destroy_process_group()


def test_bf16_wire_format_reuses_one_buffer_per_bucket():
    """GradBuckets(comm_dtype=torch.bfloat16): the gradients travel as bf16 (half the xGMI bytes, lossy) through ONE persistent
    buffer per bucket, and come back as the rank mean within bf16 rounding."""
    import socket
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_bf16_wire, args=(2, port, ret), nprocs=2, join=True)
    if ret[0][0] != 'ok':
        pytest.skip('this gloo build does not reduce bfloat16: ' + ret[0][0])
    for r in range(2):
        status, g, ptrs = ret[r]
        assert ptrs[0] == ptrs[1], 'the wire buffer is allocated once'
        # step 1: rank r feeds x = r + 2 to both rows: d/dW = 2 x per weight column; mean over ranks = 2 * 2.5 = 5
        assert torch.allclose(g, torch.full((8, 8), 5.0), rtol=1e-2), g
    assert torch.equal(ret[0][1], ret[1][1])


def test_shard_clips_partition():
    sys.path.insert(0, os.path.join(ROOT, 'videotransformer-pytorch_amd'))
    from vtx import dp
    parts = [dp.shard_clips(10, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(10))
    assert parts[1] == [1, 5, 9]
