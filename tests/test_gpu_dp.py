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


def test_bench_runs_the_rccl_path_on_one_gpu():
    """bench.py with VTX_FORCE_DP=1: the DP step (bucketed all-reduce in a 1-rank RCCL group) end to end; and
    `--gpus 2` on a one-GPU box says so and reports the real device count."""
    env = dict(os.environ, VTX_FORCE_DP='1', MASTER_PORT=str(_port()))
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '1', '--batch', '2',
           '--no-cpu-baseline', '--no-breakdown']
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
