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


class _FusedBase(torch.optim.Optimizer):
    def __init__(self, params, defaults, clip_grad=None):
        if hasattr(params, 'buckets'):                  # a vtx.dp.GradBuckets: its parameters, in bucket order
            params = [p for b in params.buckets for p in b['params']]
        super().__init__(params, defaults)
        self.clip_grad = clip_grad
        self.last_grad_norm = None                      # device scalar: ||(per-parameter norms)||_2 of the last step
        self._key = None
        self._n_steps = 0
        self._skipped = frozenset()                     # id() of parameters that take no update (see set_skipped)

    def set_skipped(self, params):
        """Parameters that received NO gradient in the step at hand.  With gradient buckets every ``p.grad`` is a (zero-filled)
        view that is never None, so "unused this step" has to be said explicitly: torch's optimizers skip ``grad is None``
        parameters altogether -- no weight decay, no momentum update -- and so does the step after this call (the reference runs
        DDP with find_unused_parameters=True for such models, model_pretrain.py:200-204)."""
        ids = frozenset(id(p) for p in params)
        if ids != self._skipped:
            self._skipped = ids
            self._key = None

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

    def _build(self, ent):
        lib = _lib.load()
        dev = ent[0][0].device
        ops.need_cuda(*[p for p, _ in ent])
        tab = (_lib.MtTensor * len(ent))()
        starts = [0]
        for i, (p, gi) in enumerate(ent):
            s1, s2 = self._state_tensors(p)
            tab[i].p, tab[i].g = p.data_ptr(), p.grad.data_ptr()
            tab[i].s1, tab[i].s2 = s1.data_ptr(), (s2.data_ptr() if s2 is not None else None)
            tab[i].n = p.numel()
            starts.append(starts[-1] + lib.vtx_mt_chunks(p.numel()))
        self._tab_host = tab
        self._groups_of = [gi for _, gi in ent]
        self._n_chunks = starts[-1]
        self._chunk_start = torch.tensor(starts, dtype=torch.int32).to(dev)
        self._tab_dev = torch.empty(C.sizeof(tab), dtype=torch.uint8, device=dev)
        self._partial = torch.empty(self._n_chunks, dtype=torch.float32, device=dev)
        self._norms = torch.zeros(len(ent) + 1, dtype=torch.float32, device=dev)
        self._hyper = None

    def _sync_tables(self):
        ent = self._entries()
        if not ent:
            return False
        key = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel()) for p, _ in ent)
        if key != self._key:
            self._build(ent)
            self._key = key
        hyper = tuple((float(g['lr']), float(g['weight_decay'])) for g in self.param_groups)
        if hyper != self._hyper:                        # schedulers rewrite lr / weight_decay between steps
            for i, gi in enumerate(self._groups_of):
                self._tab_host[i].lr, self._tab_host[i].wd = hyper[gi]
            raw = bytes(self._tab_host)
            self._tab_dev.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8), non_blocking=False)
            self._hyper = hyper
        return True

    def grad_norm(self):
        """||(||g_0||, ||g_1||, ...)||_2 over the parameters that have gradients, as a device scalar -- what
        the reference's clip_gradients returns (model_trainer.py:169)."""
        if not self._sync_tables():
            return None
        _lib.call('vtx_mt_grad_norms', self._tab_dev.data_ptr(), self._chunk_start.data_ptr(), len(self._groups_of),
                  self._n_chunks, self._partial.data_ptr(), self._norms.data_ptr(), ops.stream())
        return self._norms[-1]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._sync_tables():
            return loss
        clip = float(self.clip_grad) if self.clip_grad else 0.0
        if clip > 0.0:
            self.last_grad_norm = self.grad_norm()
        self._n_steps += 1
        self._launch(clip)
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

    def _launch(self, clip):
        raise NotImplementedError

    # ---- checkpoint surface ----------------------------------------------------------------------
    # The device table holds raw pointers to the state tensors: anything that replaces them (load_state_dict) or adds
    # parameters (add_param_group) must rebuild it.  The update count lives in the state dict the way torch keeps it
    # (a per-parameter 'step' entry), so that a resumed AdamW continues with the right bias correction and state dicts
    # move between these classes and torch.optim.SGD / AdamW.
    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._key = None

    def state_dict(self):
        sd = super().state_dict()
        # torch packs the LIVE per-parameter dicts by reference: add the 'step' entry to shallow copies, or every save would leave
        # a stale key in the optimizer's own state.  One tensor PER parameter: torch.optim.AdamW increments each parameter's
        # 'step' in place after loading such a dict -- a shared tensor would count every parameter's update.
        sd['state'] = {k: {**v, 'step': torch.tensor(float(self._n_steps))} for k, v in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = set()
        for st in self.state.values():
            if 'step' in st:
                steps.add(int(float(st.pop('step'))))
        if len(steps) > 1:
            raise NotImplementedError(f'vtx.optim: one update count for all parameters expected, the state dict holds {sorted(steps)}')
        self._n_steps = steps.pop() if steps else 0
        self._key = None                                # the loaded state tensors are new tensors


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

    def _launch(self, clip):
        g0 = self.param_groups[0]
        if any(g['momentum'] != g0['momentum'] or g['nesterov'] != g0['nesterov'] for g in self.param_groups):
            raise NotImplementedError('vtx.optim.FusedSGD: momentum / nesterov must be the same in every group')
        if any(g.get('dampening', 0) or g.get('maximize', False) for g in self.param_groups):
            raise NotImplementedError('vtx.optim.FusedSGD: dampening / maximize are not implemented')
        _lib.call('vtx_mt_sgd_step', self._tab_dev.data_ptr(), self._chunk_start.data_ptr(), len(self._groups_of),
                  self._n_chunks, self._norms.data_ptr(), clip, float(g0['momentum']), int(bool(g0['nesterov'])),
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

    def _launch(self, clip):
        g0 = self.param_groups[0]
        if any(g['betas'] != g0['betas'] or g['eps'] != g0['eps'] for g in self.param_groups):
            raise NotImplementedError('vtx.optim.FusedAdamW: betas / eps must be the same in every group')
        if any(g.get('amsgrad', False) or g.get('maximize', False) or not g.get('decoupled_weight_decay', True) for g in self.param_groups):
            raise NotImplementedError('vtx.optim.FusedAdamW: amsgrad / maximize / coupled weight decay are not implemented')
        _lib.call('vtx_mt_adamw_step', self._tab_dev.data_ptr(), self._chunk_start.data_ptr(), len(self._groups_of),
                  self._n_chunks, self._norms.data_ptr(), clip, float(g0['betas'][0]), float(g0['betas'][1]),
                  float(g0['eps']), self._n_steps, ops.stream())
