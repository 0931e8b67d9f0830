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
