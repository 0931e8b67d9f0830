"""Data-parallel gradient exchange over RCCL (torch.distributed backend "nccl" on ROCm): the bucketed
all-reduce of vtx/dp.py on real devices.  One rank always (a 1-rank RCCL communicator still runs the
collective kernels); two ranks when the box has two GPUs."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


_WORKER = r'''
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, 'videotransformer-pytorch_amd')]
import torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(rank)
dev = torch.device('cuda', rank)
dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)
from vtx import dp
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.GELU(), torch.nn.Linear(96, 32), torch.nn.Linear(32, 8)).to(dev)
dp.broadcast_parameters(model)
params = list(model.parameters())
buckets = dp.GradBuckets(params, bucket_bytes=16 << 10, force_comm=True)
assert len(buckets.buckets) >= 2
g = torch.Generator().manual_seed(5)
X = torch.randn(8, 64, generator=g).to(dev)                 # the global batch; rank r takes clips r, r+world, ...
mine = dp.shard_clips(8, rank, world)
for step in range(2):
    buckets.zero()
    model(X[mine]).square().sum().backward()                # sum over the shard
    buckets.finish()                                        # -> mean over ranks of the shard sums
got = [p.grad.clone() for p in params]
buckets.remove()                                            # detach the hooks: the next backward is the single-process reference
for p in params:
    p.grad = None
ref_model = model
ref_model(X).square().sum().backward()                      # single-process gradient of the whole batch
for a, p in zip(got, params):
    want = p.grad / world
    err = (a - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
    assert err < 1e-5, err
assert dist.get_backend() == 'nccl'
dist.barrier()
dist.destroy_process_group()
print('RANK_OK', rank, world)
'''


def _run(world):
    port = _port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, '-c', _WORKER % {'root': ROOT}], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f'RANK_OK {r} {world}' in o, o[-2000:]


def test_grad_buckets_over_rccl_one_rank():
    _run(1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_grad_buckets_over_rccl_two_ranks():
    _run(2)


_WORKER2 = r"""
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, 'videotransformer-pytorch_amd')]
import torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)                                     # BOTH ranks on the one GPU of the box
dev = torch.device('cuda', 0)
import vtx
from vtx import dp, optim, functions
import video_transformer as V
from transformer import DropPath
dp.init_process_group(dev, rank, world, backend='gloo')      # RCCL refuses two ranks on one device; gloo does not
assert dist.get_backend() == 'gloo' and dist.get_world_size() == world
vtx.set_precision('fp32')
torch.manual_seed(100 + rank)                                # a different initialisation per rank, on purpose
model = V.TimeSformer(num_frames=4, img_size=64, patch_size=16, embed_dims=128, num_heads=2, num_transformer_layers=2).to(dev).train()
for m in model.modules():
    if isinstance(m, DropPath):
        m.dropout_p = 0.0                                    # per-clip draws would differ between the sharded and the whole batch
dp.broadcast_parameters(model)                               # rank 0's weights everywhere (staged through host memory on gloo)
params = list(model.parameters())
w0 = [p.detach().clone() for p in params]
g = torch.Generator().manual_seed(7)
X = torch.randn(8, 4, 3, 64, 64, generator=g).to(dev)        # the global batch of 8 clips
mine = dp.shard_clips(8, rank, world)
buckets = dp.GradBuckets(params, bucket_bytes=512 << 10, direct=True)   # the bench / trainer configuration: kernels write the views
assert len(buckets.buckets) >= 2
opt = optim.FusedSGD(buckets, lr=0.05, momentum=0.9, nesterov=True)
buckets.zero()
model(X[mine]).square().sum().backward()                     # shard sum; finish() -> mean over ranks
buckets.finish()
got_g = [p.grad.clone() for p in params]
opt.step()
got_w = [p.detach().clone() for p in params]
buckets.remove()                                             # hooks and direct-gradient mode off: plain autograd below
# the single-process whole-batch step from the same start
with torch.no_grad():
    for p, w in zip(params, w0):
        p.copy_(w)
        p.grad = None
functions.clear_weight_cache()
(model(X).square().sum() / world).backward()
ref = torch.optim.SGD(params, lr=0.05, momentum=0.9, nesterov=True)
worst_g = max(((a - p.grad).abs().max() / p.grad.abs().max().clamp(min=1e-30)).item() for a, p in zip(got_g, params))
ref.step()
worst_w = max(((a - p.detach()).abs().max() / p.detach().abs().max().clamp(min=1e-30)).item() for a, p in zip(got_w, params))
assert all(torch.equal(a, b) for a, b in zip(w0, [w.clone() for w in w0]))
assert worst_g < 2e-5 and worst_w < 2e-6, (worst_g, worst_w)
# every rank ended with the same weights
chk = torch.stack([w.double().sum() for w in got_w]).cpu()
both = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
assert all(torch.equal(both[0], b) for b in both), 'ranks diverged'
dist.barrier()
dist.destroy_process_group()
print('RANK_OK', rank, world, 'grad %%.2e weight %%.2e' %% (worst_g, worst_w))
"""


def test_two_ranks_on_one_gpu_real_stack_over_gloo():
    """N > 1 with the PRODUCT's data-parallel stack -- a vtx TimeSformer, GradBuckets(direct=True) (the kernels accumulate
    into the bucket views and fire the hooks themselves), FusedSGD over the buckets -- as two processes sharing cuda:0 over
    gloo (buckets staged through pinned host memory): each rank steps on its shard_clips half, gradients and post-step
    weights equal the single-process whole-batch step from the same start, both ranks end identical."""
    port = _port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, '-c', _WORKER2 % {'root': ROOT}], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f'RANK_OK {r} 2' in o, o[-3000:]


def test_model_pretrain_spawns_and_joins_two_ranks(tmp_path):
    """`python model_pretrain.py ... -gpus 0 0` (two ranks; both land on device 0 of a one-GPU box) with VTX_DP_BACKEND=gloo:
    _spawn_ranks -> torch.distributed.run -> single_run on every rank -> native DP loop -> barrier -> exit code 0, rank 0's
    checkpoint on disk."""
    pkg = os.path.join(ROOT, 'videotransformer-pytorch_amd')
    argv = ['-epoch', '1', '-batch_size', '2', '-root_dir', str(tmp_path), '-num_class', '10', '-num_frames', '2', '-frame_interval', '4',
            '-train_data_path', 'synthetic', '-lr', '0.64', '-objective', 'supervised', '-img_size', '32', '-optim_type', 'sgd',
            '-synthetic_steps', '2', '-gpus', '0', '0', '-log_interval', '1']
    env = dict(os.environ, VTX_DP_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(pkg, 'model_pretrain.py')] + argv, env=env, capture_output=True, text=True,
                       timeout=900, cwd=pkg)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert 'on 2 GPU(s)' in r.stdout, r.stdout[-1500:]
    ck = [os.path.join(d, f) for d, _, fs in os.walk(str(tmp_path)) for f in fs if f == 'last_checkpoint.pth']
    assert len(ck) == 1, ck


def test_bench_runs_the_rccl_path_on_one_gpu():
    """bench.py with VTX_FORCE_DP=1: the DP step (bucketed all-reduce in a 1-rank RCCL group) end to end; and
    `--gpus 2` on a one-GPU box says so and reports the real device count."""
    env = dict(os.environ, VTX_FORCE_DP='1', MASTER_PORT=str(_port()))
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '1', '--batch', '2',
           '--no-cpu-baseline', '--no-breakdown', '--no-other-configs']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and 'RCCL all-reduce (world size 1)' in line['config']['grad_exchange']
    if torch.cuda.device_count() == 1:
        env = dict(os.environ)
        env.pop('WORLD_SIZE', None)
        r = subprocess.run(cmd[:2] + ['--gpus', '2'] + cmd[2:], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        assert 'this node has 1 GPU' in r.stderr
        assert json.loads(r.stdout.strip().splitlines()[-1])['n_gpus'] == 1
