"""Multi-tensor optimizer step + per-parameter gradient norms / clipping on libvtx kernels.

The step right after backward.  The reference builds ``torch.optim.SGD(momentum=0.9, nesterov=True)``
or ``AdamW(betas=(0.9, 0.999))`` over a no-decay / decay pair of parameter groups (optimizer.py:21-62),
clips every parameter's gradient to ``clip_grad`` by its own norm with ~250 ``torch.norm`` launches
(model_trainer.py:155-170) and rewrites the decay group's ``weight_decay`` every step
(model_trainer.py:147-151).  ``FusedSGD`` / ``FusedAdamW`` are ``torch.optim.Optimizer`` subclasses
with the same ``param_groups`` / ``state_dict`` surface (so that code keeps working) whose ``step()``
is three kernel launches over a device table of (param, grad, state) pointers: chunk sums of
squares, per-parameter norms (+ the norm of norms the reference logs), clipped update.
"""
import ctypes as C

import torch

from . import _lib, ops


class _Table:
    """One launch's device table: (param, grad, state) pointers, chunk starts, norm scratch."""


class _FusedBase(torch.optim.Optimizer):
    def __init__(self, params, defaults, clip_grad=None):
        if hasattr(params, 'buckets'):                  # a vtx.dp.GradBuckets: its parameters, in bucket order
            params = [p for b in params.buckets for p in b['params']]
        super().__init__(params, defaults)
        self.clip_grad = clip_grad
        self.last_grad_norm = None                      # device scalar: ||(per-parameter norms)||_2 of the last step
        # device tables by partition key (the data pointers of a launch's parameters / gradients): a model whose set of unused
        # parameters changes from step to step, or whose AdamW step counts split it into several launches, alternates between a
        # few partitions -- each keeps its table and device buffers instead of being rebuilt (allocations + blocking copies) on
        # every change, twice per step with clipping
        self._table_cache = {}
        self._skipped = frozenset()                     # id() of parameters that take no update (see set_skipped)

    def _step_of(self, p):
        """Number of updates ``p`` has taken: kept where torch keeps it, in ``self.state[p]['step']`` (a plain int here) -- it
        lives and dies with the parameter's state entry, whatever happens to the parameter object's id()."""
        return int(self.state[p].get('step', 0))

    def set_skipped(self, params):
        """Parameters that received NO gradient in the step at hand.  With gradient buckets every ``p.grad`` is a (zero-filled)
        view that is never None, so "unused this step" has to be said explicitly: torch's optimizers skip ``grad is None``
        parameters altogether -- no weight decay, no momentum update -- and so does the step after this call (the reference runs
        DDP with find_unused_parameters=True for such models, model_pretrain.py:200-204)."""
        self._skipped = frozenset(id(p) for p in params)      # (the tables are looked up by the entries that remain: no rebuild)

    # ---- device tables --------------------------------------------------------------------------
    def _entries(self):
        ent = []
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None or id(p) in self._skipped:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise TypeError('vtx.optim: parameters and gradients must be float32')
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise ValueError('vtx.optim: parameters and gradients must be contiguous')
                ent.append((p, gi))
        return ent

    def _state_tensors(self, p):
        raise NotImplementedError

    def _partition(self, ent):
        """Entries that one launch can update together.  SGD: all of them.  AdamW: the bias correction depends on a parameter's
        OWN update count (torch keeps state[p]['step'] per parameter), so parameters that were skipped for some steps
        (set_skipped) form their own launch -- one table in the usual case where every parameter has been updated equally often."""
        return [ent]

    def _build(self, ent):
        lib = _lib.load()
        dev = ent[0][0].device
        ops.need_cuda(*[p for p, _ in ent])
        t = _Table()
        tab = (_lib.MtTensor * len(ent))()
        starts = [0]
        for i, (p, gi) in enumerate(ent):
            s1, s2 = self._state_tensors(p)
            tab[i].p, tab[i].g = p.data_ptr(), p.grad.data_ptr()
            tab[i].s1, tab[i].s2 = s1.data_ptr(), (s2.data_ptr() if s2 is not None else None)
            tab[i].n = p.numel()
            starts.append(starts[-1] + lib.vtx_mt_chunks(p.numel()))
        t.params = [p for p, _ in ent]
        t.tab_host = tab
        t.groups_of = [gi for _, gi in ent]
        t.n_chunks = starts[-1]
        t.chunk_start = torch.tensor(starts, dtype=torch.int32).to(dev)
        t.tab_dev = torch.empty(C.sizeof(tab), dtype=torch.uint8, device=dev)
        t.partial = torch.empty(t.n_chunks, dtype=torch.float32, device=dev)
        t.norms = torch.zeros(len(ent) + 1, dtype=torch.float32, device=dev)
        t.hyper = None
        return t

    def _sync_tables(self):
        """The device tables of this step's launches: a list of _Table (empty when no parameter has a gradient)."""
        ent = self._entries()
        if not ent:
            return []
        tables = []
        for part in self._partition(ent):
            key = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel(), gi) for p, gi in part)
            t = self._table_cache.get(key)
            if t is None:
                if len(self._table_cache) >= 16:        # bounded: drop the oldest partition
                    self._table_cache.pop(next(iter(self._table_cache)))
                t = self._build(part)
                self._table_cache[key] = t
            tables.append(t)
        hyper = tuple((float(g['lr']), float(g['weight_decay'])) for g in self.param_groups)
        for t in tables:
            if hyper != t.hyper:                        # schedulers rewrite lr / weight_decay between steps
                for i, gi in enumerate(t.groups_of):
                    t.tab_host[i].lr, t.tab_host[i].wd = hyper[gi]
                raw = bytes(t.tab_host)
                t.tab_dev.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8), non_blocking=False)
                t.hyper = hyper
        return tables

    def grad_norm(self):
        """||(||g_0||, ||g_1||, ...)||_2 over the parameters that have gradients, as a device scalar -- what
        the reference's clip_gradients returns (model_trainer.py:169)."""
        tables = self._sync_tables()
        if not tables:
            return None
        for t in tables:
            _lib.call('vtx_mt_grad_norms', t.tab_dev.data_ptr(), t.chunk_start.data_ptr(), len(t.groups_of),
                      t.n_chunks, t.partial.data_ptr(), t.norms.data_ptr(), ops.stream())
        if len(tables) == 1:
            return tables[0].norms[-1]
        return torch.stack([t.norms[-1] for t in tables]).norm()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        tables = self._sync_tables()
        if not tables:
            return loss
        clip = float(self.clip_grad) if self.clip_grad else 0.0
        if clip > 0.0:
            self.last_grad_norm = self.grad_norm()
        for t in tables:
            for p in t.params:
                self.state[p]['step'] = self._step_of(p) + 1
            self._launch(t, clip, self._step_of(t.params[0]))
        self._bump_versions()
        # the staged bf16 W / W^T copies of the updated weights: one launch now instead of one per weight in the next forward
        from . import functions
        functions.restage_weights([p for g in self.param_groups for p in g['params']
                                   if p.grad is not None and p.ndim >= 2 and id(p) not in self._skipped])
        return loss

    def _bump_versions(self):
        # The kernels update parameter storage behind autograd's back; staged bf16 weight copies are
        # keyed on the version counter (vtx.functions.weights), so bump it the documented way:
        # an in-place no-op on a zero-element view costs nothing and increments the shared counter.
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    p.view(-1)[:0].zero_()

    def _launch(self, table, clip, step):
        raise NotImplementedError

    # ---- checkpoint surface ----------------------------------------------------------------------
    # The device table holds raw pointers to the state tensors: anything that replaces them (load_state_dict) or adds
    # parameters (add_param_group) must rebuild it.  The update count lives in the state dict the way torch keeps it
    # (a per-parameter 'step' entry), so that a resumed AdamW continues with the right bias correction and state dicts
    # move between these classes and torch.optim.SGD / AdamW.
    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._table_cache = {}

    def state_dict(self):
        sd = super().state_dict()
        # torch packs the LIVE per-parameter dicts by reference: add the 'step' entry to shallow copies, or every save would leave
        # a stale key in the optimizer's own state.  One tensor PER parameter: torch.optim.AdamW increments each parameter's
        # 'step' in place after loading such a dict -- a shared tensor would count every parameter's update.
        sd['state'] = {k: {**v, 'step': torch.tensor(float(v.get('step', 0)))} for k, v in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            if 'step' in st:
                st['step'] = int(float(st['step']))
        self._table_cache = {}                          # the loaded state tensors are new tensors


class FusedSGD(_FusedBase):
    """torch.optim.SGD(lr, momentum, nesterov, weight_decay; dampening 0) as one multi-tensor kernel."""

    def __init__(self, params, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=0.0, clip_grad=None):
        if nesterov and momentum <= 0:
            raise ValueError('Nesterov momentum requires a momentum')
        # the keys torch.optim.SGD keeps in its param_groups ride along unchanged (dampening 0, ...), so that a state dict
        # saved here loads into torch's class and vice versa
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=nesterov,
                                      maximize=False, foreach=None, differentiable=False, fused=None), clip_grad)

    def _state_tensors(self, p):
        st = self.state[p]
        if 'momentum_buffer' not in st:
            # zero-initialised: the first update gives buf = momentum * 0 + g = g, torch's first-step rule
            st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st['momentum_buffer'], None

    def _launch(self, t, clip, step):
        g0 = self.param_groups[0]
        if any(g['momentum'] != g0['momentum'] or g['nesterov'] != g0['nesterov'] for g in self.param_groups):
            raise NotImplementedError('vtx.optim.FusedSGD: momentum / nesterov must be the same in every group')
        if any(g.get('dampening', 0) or g.get('maximize', False) for g in self.param_groups):
            raise NotImplementedError('vtx.optim.FusedSGD: dampening / maximize are not implemented')
        _lib.call('vtx_mt_sgd_step', t.tab_dev.data_ptr(), t.chunk_start.data_ptr(), len(t.groups_of),
                  t.n_chunks, t.norms.data_ptr(), clip, float(g0['momentum']), int(bool(g0['nesterov'])),
                  0, ops.stream())


class FusedAdamW(_FusedBase):
    """torch.optim.AdamW(lr, betas, eps, weight_decay) as one multi-tensor kernel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, clip_grad=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=True), clip_grad)    # torch.optim.AdamW's keys (see FusedSGD)

    def _state_tensors(self, p):
        st = self.state[p]
        if 'exp_avg' not in st:
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st['exp_avg'], st['exp_avg_sq']

    def _partition(self, ent):
        by_step = {}
        for e in ent:
            by_step.setdefault(self._step_of(e[0]), []).append(e)
        return [by_step[k] for k in sorted(by_step, reverse=True)]

    def _launch(self, t, clip, step):
        g0 = self.param_groups[0]
        if any(g['betas'] != g0['betas'] or g['eps'] != g0['eps'] for g in self.param_groups):
            raise NotImplementedError('vtx.optim.FusedAdamW: betas / eps must be the same in every group')
        if any(g.get('amsgrad', False) or g.get('maximize', False) or not g.get('decoupled_weight_decay', True) for g in self.param_groups):
            raise NotImplementedError('vtx.optim.FusedAdamW: amsgrad / maximize / coupled weight decay are not implemented')
        _lib.call('vtx_mt_adamw_step', t.tab_dev.data_ptr(), t.chunk_start.data_ptr(), len(t.groups_of),
                  t.n_chunks, t.norms.data_ptr(), clip, float(g0['betas'][0]), float(g0['betas'][1]),
                  float(g0['eps']), step, ops.stream())
