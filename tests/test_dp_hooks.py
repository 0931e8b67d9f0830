"""vtx.dp.GradBuckets hook accounting with direct gradients: the kernels call a parameter's hooks themselves
(vtx.functions._fire) and torch runs them once more for the None the Function returns -- that duplicate is ignored; a
SECOND kernel accumulation into an already counted parameter (a module applied twice in one backward) raises instead
of racing the bucket's all-reduce."""
import pytest
import torch


def test_direct_mode_counts_a_parameter_once_and_rejects_a_second_accumulation():
    from vtx import dp, functions
    ps = [torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(3))]
    gb = dp.GradBuckets(ps, direct=True)
    try:
        gb.zero()
        hook = next(iter(ps[0]._post_accumulate_grad_hooks.values()))
        functions._fire(ps[0])                      # the kernel's own call
        hook(ps[0])                                 # autograd's duplicate for the returned None: ignored
        assert sum(b['pending'] for b in gb.buckets) == 1
        with pytest.raises(RuntimeError, match='second kernel accumulated'):
            functions._fire(ps[0])
        assert not functions.firing()
        gb.zero()
        functions._fire(ps[0], ps[1])
        assert sum(b['pending'] for b in gb.buckets) == 0
    finally:
        gb.remove()
    assert not functions.direct_grads_enabled()
    gb2 = dp.GradBuckets(ps, direct=False)
    try:
        gb2.zero()
        hook = next(iter(ps[0]._post_accumulate_grad_hooks.values()))
        hook(ps[0])
        with pytest.raises(RuntimeError, match='second gradient arrived'):
            hook(ps[0])
    finally:
        gb2.remove()


def test_one_rank_group_retries_another_port_when_the_probed_one_is_taken(monkeypatch):
    """vtx.dp.init_process_group: a one-rank group (VTX_FORCE_DP) owns its rendezvous port; "free when probed" is not "free
    when the store binds" (seen once as EADDRINUSE in the GPU suite) -> another port, not an error.  More than one rank: the
    launcher's port, one attempt."""
    import os
    from vtx import dp
    calls = []

    def fake_init(backend, device_id=None, rank=0, world_size=1):
        calls.append((os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'], rank, world_size))
        if world_size == 1 and len(calls) < 3:
            raise RuntimeError('The server socket has failed to listen on any local network address. port: %s, useIpv6: false, '
                               'code: -98, name: EADDRINUSE, message: address already in use' % os.environ['MASTER_PORT'])
    monkeypatch.setattr(dp.dist, 'init_process_group', fake_init)
    monkeypatch.setenv('MASTER_PORT', '1')
    dp.init_process_group(torch.device('cpu'), 0, 1)
    assert len(calls) == 3 and all(c[0] == '127.0.0.1' and c[2:] == (0, 1) for c in calls)
    assert all(c[1] != '1' for c in calls)                       # the inherited port is not used by a one-rank group
    calls.clear()
    monkeypatch.setenv('MASTER_PORT', '23456')
    dp.init_process_group(torch.device('cpu'), 3, 8)
    assert calls == [('127.0.0.1', '23456', 3, 8)]

    def other_error(backend, device_id=None, rank=0, world_size=1):
        raise RuntimeError('no RCCL here')
    monkeypatch.setattr(dp.dist, 'init_process_group', other_error)
    with pytest.raises(RuntimeError, match='no RCCL'):
        dp.init_process_group(torch.device('cpu'), 0, 1)
