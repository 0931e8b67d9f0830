"""Data-parallel gradient exchange for the clip-sharded training step.

The reference gets this implicitly from Lightning's ``accelerator="ddp"``
(model_pretrain.py:200-204): clips are independent, every rank holds a full replica,
and the only exchange per step is a mean all-reduce of the parameter gradients.

Here: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL on ROCm) over
xGMI.  Gradients live in a few flat fp32 buffers (``param.grad`` are views), bucketed
per transformer layer in reverse registration order = the order backward produces them.
A bucket's all-reduce is issued asynchronously from the post-accumulate hook of its last
parameter, so the exchange of layer i+1 overlaps the backward kernels of layer i; the
step ends with ``finish()`` (wait + 1/world scaling folded into the reduce op).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of S bytes is
bound by 2*(7/8)*S / link rate, so buckets are large (one layer, ~40 MB fp32) rather
than torch DDP's NVSwitch-era 25 MB default.
"""
import os

import torch
import torch.distributed as dist


def _host_side(group, t):
    """True when the group's backend cannot take a device tensor as it is: anything but RCCL (gloo: the two-ranks-on-one-GPU
    tests, a debug rendezvous without xGMI).  Such collectives are staged through pinned host memory."""
    return t.is_cuda and dist.get_backend(group) != 'nccl'


def all_reduce_sum(t, group=None):
    """Blocking sum all-reduce of a (device) tensor on whatever backend the group has."""
    if _host_side(group, t):
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()      # h is freed on return
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class GradBuckets:
    def __init__(self, params, bucket_bytes=48 << 20, process_group=None, comm_dtype=None, force_comm=False,
                 direct=False, static_unused=False):
        """direct=True: the block kernels accumulate parameter gradients straight into the bucket views (no
        per-parameter `grad += new` kernels; see vtx.functions.set_direct_grads) -- a process-wide switch that
        stays on until set_direct_grads(False) or remove().
        static_unused=True: the parameters `unfired()` reports (no gradient on ANY rank) are taken to be unused in every later
        step as well and no longer hold their bucket back: buckets go out in order, so ONE parameter without a gradient -- the
        tail of a pretrain model such as MaskFeat's unused head -- otherwise keeps its bucket and every later one until
        finish(), i.e. the whole exchange loses its overlap with backward.  Should such a parameter get a gradient
        after all, it is counted again from the next step on; if its bucket has already gone out this step, the hook raises."""
        self.group = process_group
        self.static_unused = bool(static_unused)
        self._known_unused = set()               # id() of parameters every rank agreed are unused (static_unused)
        self._warned_held = False
        self.direct = bool(direct)
        if self.direct:
            from . import functions
            functions.set_direct_grads(True)
        self.force_comm = force_comm            # issue the collectives even in a 1-rank group (exercises RCCL)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm_dtype = comm_dtype            # e.g. torch.bfloat16 to halve xGMI bytes (lossy)
        params = [p for p in params if p.requires_grad]
        order = list(reversed(params))
        self.buckets = []                       # each: dict(params, flat, pending, handle)
        cur, cur_bytes = [], 0
        for p in order:
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._hooks = []
        self._fired = set()                     # id() of the parameters whose gradient has arrived since zero()
        self._unfired_cache = None
        for bi, b in enumerate(self.buckets):
            for p in b['params']:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self._launched = []
        self._next = 0                          # the next bucket to go out (see _launch_ready)

    def _close(self, plist):
        # every gradient view starts on a 16-byte boundary (the kernels that write gradients in place take 16-byte vectors):
        # a parameter whose element count is not a multiple of 4 (a 10-class head bias) is followed by a few unused floats
        pad4 = lambda n: (n + 3) // 4 * 4               # noqa: E731
        n = sum(pad4(p.numel()) for p in plist)
        flat = torch.zeros(n, dtype=torch.float32, device=plist[0].device)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += pad4(p.numel())
        self.buckets.append(dict(params=plist, flat=flat, pending=len(plist), handle=None, comm=None))

    def _make_hook(self, bi):
        def hook(p):
            b = self.buckets[bi]
            if id(p) in self._known_unused:
                # a parameter all ranks had agreed is unused gets a gradient after all: it was not counted in `pending`
                if bi < self._next:
                    raise RuntimeError('GradBuckets(static_unused=True): a parameter that was unused in earlier steps received a '
                                       'gradient after the all-reduce of its bucket had been issued -- the gradient would be lost; '
                                       'build the buckets with static_unused=False for models whose graph changes between steps')
                self._known_unused.discard(id(p))
                self._fired.add(id(p))
                return
            if id(p) in self._fired:
                # direct gradients: the kernels' own call (vtx.functions._fire) came first; this torch version also
                # runs a parameter's post-accumulate hooks when the Function returned None for it -- count once
                if self.direct:
                    from . import functions
                    if functions.firing():
                        raise RuntimeError('GradBuckets(direct=True): a second kernel accumulated into a parameter whose '
                                           'gradient was already counted (a module applied twice in one backward); its '
                                           "bucket's all-reduce may already be running -- use direct=False for such models")
                    return
                raise RuntimeError('GradBuckets: a second gradient arrived for a parameter whose bucket was already '
                                   'counted -- call zero() before every backward (one backward per step)')
            self._fired.add(id(p))
            b['pending'] -= 1
            if b['pending'] == 0:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Collectives are issued strictly in bucket order: bucket k goes out when its last gradient has arrived AND buckets
        0 .. k-1 have gone out.  Every rank must issue the same sequence of all-reduces; a rank on which some parameter stays
        unused this step (a data-dependent branch) completes its buckets in another order than its peers, and issuing them in
        completion order would pair different buckets across ranks (found by tests/test_dp_gloo.py::
        test_unfired_parameters_are_agreed_across_ranks; torch DDP orders its buckets the same way).  With every parameter used,
        bucket order IS completion order (reverse registration order = the order backward produces gradients)."""
        while self._next < len(self.buckets) and self.buckets[self._next]['pending'] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b):
        if self.world == 1 and not self.force_comm:
            return
        buf = b['flat']
        if self.comm_dtype is not None and self.comm_dtype != buf.dtype:
            # a persistent buffer per bucket (one cast launch, no allocation per step; VERDICT r5: the lossy wire format used to
            # allocate and cast through a new tensor every step)
            if b.get('comm_buf') is None:
                b['comm_buf'] = torch.empty(buf.shape, dtype=self.comm_dtype, device=buf.device)
            b['comm_buf'].copy_(buf)
            b['comm'] = b['comm_buf']
            buf = b['comm']
        op = dist.ReduceOp.AVG if dist.get_backend(self.group) == 'nccl' else dist.ReduceOp.SUM
        b['host'] = None
        if _host_side(self.group, buf):
            # not RCCL: the bucket travels through pinned host memory.  The kernels that wrote it are on the current stream:
            # the copy follows them there, and the host waits for the copy before the backend's threads read the buffer.
            if b.get('pinned') is None or b['pinned'].shape != buf.shape or b['pinned'].dtype != buf.dtype:
                b['pinned'] = torch.empty(buf.shape, dtype=buf.dtype, pin_memory=True)
            b['host'] = b['pinned']
            b['host'].copy_(buf, non_blocking=True)
            torch.cuda.current_stream(buf.device).synchronize()
            buf = b['host']
        b['handle'] = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
        b['avg_in_op'] = op == dist.ReduceOp.AVG
        self._launched.append(b)

    def zero(self):
        """Zero the flat gradient buffers (use instead of optimizer.zero_grad(set_to_none=True):
        the .grad views must stay bound to the buckets)."""
        for b in self.buckets:
            b['flat'].zero_()
            b['pending'] = sum(1 for p in b['params'] if id(p) not in self._known_unused)
            b['handle'] = None
            b['comm'] = None
        self._launched = []
        self._fired = set()
        self._unfired_cache = None
        self._next = 0

    def finish(self):
        """Wait for every bucket's all-reduce; afterwards param.grad holds the mean gradient."""
        # buckets that have not gone out yet (a parameter without a gradient this step holds its bucket -- and every later
        # one -- back): now, in bucket order, so that every rank issues the same sequence
        held = len(self.buckets) - self._next
        if held > 1 and (self.world > 1 or self.force_comm) and not self._warned_held:
            # more than the one bucket backward has just completed: some parameter got no gradient this step and held its
            # bucket -- and every later one -- back, so their all-reduces start only now, with no backward left to hide them
            import sys
            first = self.buckets[self._next]
            names = sum(1 for p in first['params'] if id(p) not in self._fired)
            sys.stderr.write(f'vtx.dp.GradBuckets: finish() had to issue {held} of {len(self.buckets)} all-reduces after backward '
                             f'(bucket {self._next} waited for {names} parameter(s) without a gradient): no overlap with backward '
                             'for them.  If those parameters are unused in every step, build the buckets with static_unused=True '
                             'and call unfired() after finish() (reported once).\n')
            self._warned_held = True
        while self._next < len(self.buckets):
            self._launch(self.buckets[self._next])
            self._next += 1
        for b in self._launched:
            b['handle'].wait()
            if b.get('host') is not None:
                (b['comm'] if b['comm'] is not None else b['flat']).copy_(b['host'], non_blocking=True)
            if b['comm'] is not None:
                b['flat'].copy_(b['comm'])
            if not b['avg_in_op']:
                b['flat'].div_(self.world)
        self._launched = []

    def unfired(self):
        """The parameters whose gradient has NOT arrived since zero() ON ANY RANK: unused in this step's graph everywhere (their
        mean gradient is zero).  COLLECTIVE when the group has more than one rank -- every rank calls it once per step, after
        finish(): the per-parameter "fired" bitmap is all-reduced with MAX, the way DDP's find_unused_parameters=True shares its
        used-parameter bitmap, so that a parameter used on one rank only (a data-dependent branch) is updated with the averaged
        gradient on EVERY rank -- skipping it on the ranks that did not use it would let the replicas drift apart silently
        (ADVICE r4).  The result is cached until the next zero()."""
        if self._unfired_cache is not None:
            return self._unfired_cache
        params = [p for b in self.buckets for p in b['params']]
        fired = [1 if id(p) in self._fired else 0 for p in params]
        if self.world > 1:
            flags = torch.tensor(fired, dtype=torch.int32)
            if dist.get_backend(self.group) == 'nccl':
                flags = flags.to(params[0].device)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
            fired = flags.cpu().tolist()
        self._unfired_cache = [p for p, f in zip(params, fired) if not f]
        if self.static_unused:                   # agreed across ranks: the same set everywhere
            self._known_unused = {id(p) for p in self._unfired_cache}
        return self._unfired_cache

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.direct:
            from . import functions
            functions.set_direct_grads(False)
            self.direct = False


def broadcast_parameters(module, src=0, group=None):
    """One-time rank-0 -> all parameter broadcast (what DDP does at wrap time)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if _host_side(group, t):
                h = t.detach().cpu()
                dist.broadcast(h, src=src, group=group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src, group=group)
    from . import functions
    functions.clear_weight_cache()          # the staged bf16 W / W^T copies of the old values are stale


def shard_clips(global_batch, rank, world):
    """Clips are the independent units: rank r takes clips r, r+world, ... of the global batch."""
    return list(range(rank, global_batch, world))



def init_process_group(device, rank=0, world=1, attempts=8, backend=None):
    """torch.distributed (RCCL) process group for one rank per GPU on a single node.  With more than one rank the launcher's
    MASTER_ADDR / MASTER_PORT are used as they are.  ``backend`` (default: $VTX_DP_BACKEND, else "nccl" = RCCL): "gloo" runs
    the same data-parallel stack with the buckets staged through pinned host memory -- RCCL refuses two ranks on one device,
    gloo does not, which is how the N > 1 path is tested on a one-GPU box (tests/test_gpu_dp.py).  A ONE-rank group (VTX_FORCE_DP=1: the DP code path on one GPU) needs no
    agreement with anybody about the port: it takes a free one itself and, because "free when probed" is not "free when the
    store binds it" (RCCL bootstrap sockets of an earlier group, another process), retries with another port on EADDRINUSE."""
    import socket
    backend = backend or os.environ.get('VTX_DP_BACKEND', 'nccl')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend != 'nccl':
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(backend, rank=rank, world_size=world)
        return
    if world > 1:
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', device_id=device, rank=rank, world_size=world)
        return
    last = None
    for _ in range(attempts):
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        os.environ['MASTER_PORT'] = str(port)
        try:
            dist.init_process_group('nccl', device_id=device, rank=0, world_size=1)
            return
        except Exception as e:                       # DistNetworkError (EADDRINUSE) from the TCP store: another port
            if 'EADDRINUSE' not in str(e) and 'address already in use' not in str(e).lower():
                raise
            last = e
    raise last
