"""Multi-tensor optimizer step and per-parameter gradient norms / clipping (csrc/optim.hip) against
torch.optim and the reference's clip_gradients arithmetic (model_trainer.py:155-170)."""
import pytest
import torch

from helpers import check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SHAPES = [(768, 768), (3,), (4097,), (2304, 768), (1, 1, 768), (5, 7, 11), (8192,), (12288 + 4,)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) for s in SHAPES]


def _reference_clip(grads, clip):
    """model_trainer.py:155-170 on CPU doubles -> (clipped grads, total norm)."""
    norms, out = [], []
    for g in grads:
        n = g.double().norm(2)
        norms.append(n)
        c = clip / (n + 1e-6) if clip else 1.0
        out.append(g.double() * c if (clip and c < 1) else g.double())
    return out, torch.stack(norms).norm(2)


@pytest.mark.parametrize('clip', [None, 0.5])
@pytest.mark.parametrize('kind', ['sgd', 'adamw'])
def test_fused_step_matches_torch(kind, clip):
    from vtx import optim
    init = _params(1)
    p_ref = [torch.nn.Parameter(t.clone().double()) for t in init]
    p_gpu = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    groups = lambda ps: [{'params': ps[:3], 'weight_decay': 0.0}, {'params': ps[3:]}]   # noqa: E731
    if kind == 'sgd':
        o_ref = torch.optim.SGD(groups(p_ref), lr=0.05, momentum=0.9, nesterov=True, weight_decay=0.05)
        o_gpu = optim.FusedSGD(groups(p_gpu), lr=0.05, momentum=0.9, nesterov=True, weight_decay=0.05, clip_grad=clip)
    else:
        o_ref = torch.optim.AdamW(groups(p_ref), lr=0.01, betas=(0.9, 0.999), weight_decay=0.05)
        o_gpu = optim.FusedAdamW(groups(p_gpu), lr=0.01, betas=(0.9, 0.999), weight_decay=0.05, clip_grad=clip)
    for step in range(4):
        grads = [g * (0.3 if i % 2 else 3.0) for i, g in enumerate(_params(10 + step))]
        clipped, total = _reference_clip(grads, clip)
        for pr, pg, g, gc in zip(p_ref, p_gpu, grads, clipped):
            pr.grad = gc.clone()
            pg.grad = g.clone().to(DEV)
        if step == 2:                                   # a scheduler rewrites the decayed group between steps
            o_ref.param_groups[1]['weight_decay'] = 0.02
            o_gpu.param_groups[1]['weight_decay'] = 0.02
            o_ref.param_groups[0]['lr'] *= 0.5
            o_gpu.param_groups[0]['lr'] *= 0.5
        v0 = p_gpu[0]._version
        o_ref.step()
        o_gpu.step()
        assert p_gpu[0]._version > v0, 'the in-place kernel update must bump the version counter (weight cache)'
        if clip:
            check(f'{kind} total grad norm step {step}', o_gpu.last_grad_norm.cpu(), total, 1e-5)
        for i, (pr, pg) in enumerate(zip(p_ref, p_gpu)):
            check(f'{kind} clip={clip} step {step} param {i}', pg.detach().cpu(), pr.detach(), 2e-6)


def test_unused_parameters_take_no_update():
    """With gradient buckets every .grad is a zero-filled view, never None: a parameter outside the step's graph is named
    through set_skipped() and then behaves like torch's `grad is None` -- no weight decay, no moment update."""
    from vtx import optim
    init = _params(3)
    p_ref = [torch.nn.Parameter(t.clone().double()) for t in init]
    p_gpu = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    o_ref = torch.optim.AdamW(p_ref, lr=0.01, weight_decay=0.1)
    o_gpu = optim.FusedAdamW(p_gpu, lr=0.01, weight_decay=0.1)
    unused = {1, 4}
    for step in range(5):
        grads = _params(20 + step)
        for i, (pr, pg, g) in enumerate(zip(p_ref, p_gpu, grads)):
            pr.grad = None if i in unused else g.clone().double()
            pg.grad = torch.zeros_like(pg) if i in unused else g.clone().to(DEV)     # what a bucket view looks like
        o_gpu.set_skipped([p_gpu[i] for i in unused])
        o_ref.step()
        o_gpu.step()
        if step == 1:
            unused = {4}                                # parameter 1 joins the graph later: its first update is step 1 of ITS history
    check('unused parameter stays put', p_gpu[4].detach().cpu(), init[4], 0.0 + 1e-12)
    # parameter 1 was skipped twice and updated three times: torch's per-parameter step count gives its bias correction for update
    # 1, 2, 3 -- the fused class keeps per-parameter counts too and launches it on its own table
    for i in (0, 1, 2, 3, 5, 6, 7):
        check(f'used parameter {i}', p_gpu[i].detach().cpu(), p_ref[i].detach(), 2e-6)
    steps = {i: float(st['step']) for i, st in o_gpu.state_dict()['state'].items()}
    assert steps[0] == 5.0 and steps[1] == 3.0 and 4 not in steps, steps
    assert {i: float(st['step']) for i, st in o_ref.state_dict()['state'].items()} == steps
    sd = o_gpu.state_dict()
    # round 6: the update count lives where torch keeps it, in the parameter's own state entry (a plain int in the live
    # state: it survives whatever happens to id(parameter)); state_dict() hands it out as one tensor per parameter
    assert all(isinstance(st['step'], int) for st in o_gpu.state.values() if 'step' in st)
    assert all(torch.is_tensor(st['step']) for st in sd['state'].values())
    assert o_gpu.state[p_gpu[1]]['step'] == 3 and 'step' not in o_gpu.state.get(p_gpu[4], {})


@pytest.mark.parametrize('kind', ['sgd', 'adamw'])
def test_checkpoint_resume_matches_torch(kind):
    """Save after two updates, load into FRESH optimizers (the fused one and torch's), continue: the AdamW bias correction
    continues from update 3 (the update count travels in the state dict the way torch keeps it), the device table
    follows the loaded state tensors, and state dicts move between the fused classes and torch.optim in both
    directions (a reference checkpoint resumes here and vice versa)."""
    from vtx import optim
    init = _params(2)
    mk_ref = (lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=0.05)) if kind == 'sgd' else \
        (lambda ps: torch.optim.AdamW(ps, lr=0.01, betas=(0.9, 0.999), weight_decay=0.05))
    mk_gpu = (lambda ps: optim.FusedSGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=0.05)) if kind == 'sgd' else \
        (lambda ps: optim.FusedAdamW(ps, lr=0.01, betas=(0.9, 0.999), weight_decay=0.05))
    p_ref = [torch.nn.Parameter(t.clone().double()) for t in init]
    p_gpu = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    o_ref, o_gpu = mk_ref(p_ref), mk_gpu(p_gpu)

    def run(o_r, o_g, steps, first):
        for step in range(first, first + steps):
            grads = _params(20 + step)
            for pr, pg, g in zip(p_ref, p_gpu, grads):
                pr.grad = g.clone().double()
                pg.grad = g.clone().to(DEV)
            o_r.step()
            o_g.step()

    run(o_ref, o_gpu, 2, 0)
    sd_ref, sd_gpu = o_ref.state_dict(), o_gpu.state_dict()
    assert all(float(st['step']) == 2.0 for st in sd_gpu['state'].values())
    # fresh optimizers over the same parameters; the fused one resumes from ITS state dict, torch from torch's
    o_ref2, o_gpu2 = mk_ref(p_ref), mk_gpu(p_gpu)
    o_ref2.load_state_dict(sd_ref)
    o_gpu2.load_state_dict(sd_gpu)
    run(o_ref2, o_gpu2, 2, 2)
    for i, (pr, pg) in enumerate(zip(p_ref, p_gpu)):
        check(f'{kind} resumed param {i}', pg.detach().cpu(), pr.detach(), 2e-6)
    # cross-loading: torch's state dict (float64 here -> cast by load_state_dict) into the fused class, and back
    o_gpu3 = mk_gpu(p_gpu)
    o_gpu3.load_state_dict(o_ref2.state_dict())
    o_ref3 = mk_ref(p_ref)
    o_ref3.load_state_dict(o_gpu2.state_dict())
    run(o_ref3, o_gpu3, 1, 4)
    for i, (pr, pg) in enumerate(zip(p_ref, p_gpu)):
        check(f'{kind} cross-loaded param {i}', pg.detach().cpu(), pr.detach(), 2e-6)
    # a step after load_state_dict must not touch the OLD state tensors (the device table was rebuilt)
    old_state = [t for st in o_gpu2.state.values() for t in st.values() if torch.is_tensor(t)]
    snap = [t.clone() for t in old_state]
    import copy
    o_gpu2.load_state_dict(copy.deepcopy(o_gpu2.state_dict()))      # new state tensors
    run(mk_ref(p_ref), o_gpu2, 1, 5)
    assert all(torch.equal(a, b) for a, b in zip(old_state, snap)), 'the kernel wrote through stale state pointers'


def test_grad_norm_is_the_reference_statistic():
    from vtx import optim
    ps = [torch.nn.Parameter(t.to(DEV)) for t in _params(3)]
    grads = _params(4)
    for p, g in zip(ps, grads):
        p.grad = g.to(DEV)
    o = optim.FusedSGD(ps, lr=0.1)
    _, total = _reference_clip(grads, None)
    check('grad_norm', o.grad_norm().cpu(), total, 1e-6)
    n1 = o.grad_norm().clone()
    assert torch.equal(n1, o.grad_norm()), 'norms are deterministic (fixed summation order)'


def test_build_optimizer_groups_like_the_reference():
    """optimizer.build_optimizer: no-decay group first (1-D, biases, pos_embed / cls_token), then the decayed one."""
    import types
    import optimizer
    import video_transformer as V
    m = V.TimeSformer(num_frames=2, img_size=32, patch_size=16, embed_dims=64, num_heads=1, num_transformer_layers=1).to(DEV)
    hp = types.SimpleNamespace(optim_type='adamw', lr=1e-3, weight_decay=0.05, arch='timesformer', layer_decay=1)
    opt = optimizer.build_optimizer(hp, m, is_pretrain=False)
    names = {id(p): n for n, p in m.named_parameters()}
    g0 = {names[id(p)] for p in opt.param_groups[0]['params']}
    g1 = {names[id(p)] for p in opt.param_groups[1]['params']}
    assert opt.param_groups[0]['weight_decay'] == 0 and opt.param_groups[1]['weight_decay'] == 0.05
    assert {'pos_embed', 'cls_token', 'norm.weight', 'patch_embed.projection.bias'} <= g0
    # time_embed is 3-D and not among the reference's no-decay keywords: it IS decayed (optimizer.py:51-58)
    assert 'patch_embed.projection.weight' in g1 and all(n.endswith('weight') or n == 'time_embed' for n in g1)
    assert len(g0) + len(g1) == len(names)


def test_restaged_weights_after_fused_step_match_a_fresh_cast():
    """The fused optimizers refresh the staged bf16 / fp32 W and W^T copies of the weights they have just updated in
    one multi-tensor launch (vtx.functions.restage_weights): bit-identical to the per-weight staging, for shapes
    that take the vector path and shapes that do not, and the next forward uses the updated weights."""
    import vtx
    from vtx import functions, ops, optim
    dev = 'cuda:0'
    torch.manual_seed(0)
    shapes = [(768, 768), (2304, 768), (96, 200), (130, 77), (64, 64), (8, 3072)]
    params = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    for dtype in (torch.bfloat16, torch.float32):
        functions.clear_weight_cache()
        staged = [functions.weights(p, dtype, True) for p in params]
        opt = optim.FusedSGD(params, lr=0.1, momentum=0.9, nesterov=True)
        for p in params:
            p.grad = torch.randn_like(p)
        before = [p.detach().clone() for p in params]
        opt.step()
        torch.cuda.synchronize()
        for p, b, (wc, wt) in zip(params, before, staged):
            assert not torch.equal(p.detach(), b)
            wc2, wt2 = functions.weights(p, dtype, True)
            assert wc2 is wc or dtype == torch.float32, 'the staged copy must have been refreshed in place (cache hit)'
            assert wt2 is wt
            rc, rt = ops.cast_transpose(p.detach(), dtype)
            assert torch.equal(wc2, rc) and torch.equal(wt2, rt), (tuple(p.shape), dtype)
    functions.clear_weight_cache()
