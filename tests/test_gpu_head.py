"""On-device Mixup / CutMix + soft targets, softmax cross-entropy and top-k accuracy (csrc/head.hip) against the
reference's ATen arithmetic (mixup.py:16-126, model_trainer.py:85-91,207-215)."""
import numpy as np
import pytest
import torch

from helpers import check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_mixup_and_cutmix_are_bit_identical_to_the_reference_arithmetic():
    from vtx import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 4 * 3, 20, 28, generator=g)
    for lam in (0.3141592, 0.9, 0.5):
        ref = x.clone()
        flipped = ref.flip(0).mul_(1. - lam)                 # mixup.py:112-113
        ref.mul_(lam).add_(flipped)
        got = ops.mixup_batch_(x.clone().to(DEV), lam).cpu()
        assert torch.equal(got, ref), f'mixup lam={lam}'
    for (yl, yh, xl, xh) in ((3, 11, 5, 20), (0, 20, 0, 28), (7, 7, 1, 2), (19, 20, 27, 28)):
        ref = x.clone()
        ref[:, :, yl:yh, xl:xh] = ref.flip(0)[:, :, yl:yh, xl:xh]     # mixup.py:110
        got = ops.cutmix_batch_(x.clone().to(DEV), yl, yh, xl, xh).cpu()
        assert torch.equal(got, ref), (yl, yh, xl, xh)


def test_mixup_target_matches_the_reference_formula():
    from vtx import ops
    labels = torch.tensor([3, 0, 9, 9, 1, 7])
    for lam, smoothing in ((0.37, 0.1), (1.0, 0.0), (0.8123, 0.2)):
        C = 10
        off = smoothing / C
        on = 1. - smoothing + off
        oh = lambda t: torch.full((t.numel(), C), off).scatter_(1, t.view(-1, 1), on)   # noqa: E731  (mixup.py:16-18)
        ref = oh(labels) * lam + oh(labels.flip(0)) * (1. - lam)
        got = ops.mixup_target(labels.to(DEV), C, lam, smoothing).cpu()
        assert torch.equal(got, ref), (lam, smoothing)


def test_mixup_class_applies_its_plan_like_the_reference_arithmetic():
    """mixup.Mixup on CUDA tensors: the plan comes from ``draw`` (host half; compared with the reference's class draw for
    draw in tests/test_reference_trainer.py), the clips and targets equal the reference's ATen formulas (mixup.py:16-25,
    :105-113) applied to that plan on the CPU, bit for bit; CPU tensors raise."""
    import mixup
    fn = mixup.Mixup(num_classes=12)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(4, 2, 3, 16, 16, generator=g)
    y0 = torch.tensor([1, 5, 5, 11])
    kinds = set()
    for seed in range(8):                                    # both mixup and cutmix plans occur
        np.random.seed(seed)
        lam, box = fn.draw((4, 6, 16, 16))
        ref = x0.clone().view(4, 6, 16, 16)
        if box is not None:
            yl, yh, xl, xh = box
            ref[:, :, yl:yh, xl:xh] = ref.flip(0)[:, :, yl:yh, xl:xh]
        elif lam != 1.:
            flipped = ref.flip(0).mul_(1. - lam)
            ref.mul_(lam).add_(flipped)
        off = 0.1 / 12
        oh = lambda t: torch.full((t.numel(), 12), off).scatter_(1, t.view(-1, 1), 1. - 0.1 + off)   # noqa: E731
        yref = oh(y0) * lam + oh(y0.flip(0)) * (1. - lam)
        np.random.seed(seed)
        xin = x0.clone().to(DEV)
        xg, yg = fn(xin, y0.to(DEV))
        assert xg.data_ptr() == xin.data_ptr(), 'mixed in place'
        assert torch.equal(xg.cpu(), ref.view(4, 2, 3, 16, 16)) and torch.equal(yg.cpu(), yref), seed
        kinds.add(box is not None)
    assert kinds == {True, False}
    assert torch.equal(mixup.one_hot(y0.to(DEV), 12, 0.7, 0.1).cpu(), torch.full((4, 12), 0.1).scatter_(1, y0.view(-1, 1), 0.7))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        fn(x0.clone(), y0)


@pytest.mark.parametrize('B,C', [(64, 400), (5, 174), (3, 7)])
def test_softmax_cross_entropy_fwd_bwd(B, C):
    from vtx import functions as F_
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(B, C, generator=g) * 3
    labels = torch.randint(0, C, (B,), generator=g)
    soft = torch.rand(B, C, generator=g)
    soft = soft / soft.sum(-1, keepdim=True)
    for name, target in (('labels', labels), ('soft', soft)):
        x64 = logits.double().requires_grad_(True)
        if name == 'labels':
            ref = torch.nn.functional.cross_entropy(x64, labels)
        else:
            ref = torch.sum(-soft.double() * torch.log_softmax(x64, dim=-1), dim=-1).mean()      # timm SoftTargetCrossEntropy
        (ref * 1.7).backward()
        xg = logits.to(DEV).requires_grad_(True)
        loss = F_.SoftmaxXentFn.apply(xg, target.to(DEV))
        (loss * 1.7).backward()
        check(f'xent {name} {B}x{C} loss', loss.detach().cpu(), ref.detach(), 1e-6)
        check(f'xent {name} {B}x{C} dlogits', xg.grad.cpu(), x64.grad, 1e-5)


def test_softmax_cross_entropy_ignored_labels():
    """nn.CrossEntropyLoss() (reference model_trainer.py:91) has ignore_index = -100: such rows give no loss, no gradient and
    are left out of the mean's denominator.  The kernels decide that on the device (they used to index out of
    bounds); top-k accuracy never counts such a row as correct."""
    from vtx import functions as F_, ops
    g = torch.Generator().manual_seed(5)
    B, C = 13, 31
    logits = torch.randn(B, C, generator=g) * 2
    labels = torch.randint(0, C, (B,), generator=g)
    labels[[1, 6, 12]] = -100
    x64 = logits.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(x64, labels)
    (ref * 0.6).backward()
    xg = logits.to(DEV).requires_grad_(True)
    loss = F_.SoftmaxXentFn.apply(xg, labels.to(DEV))
    (loss * 0.6).backward()
    check('xent ignore_index loss', loss.detach().cpu(), ref.detach(), 1e-6)
    check('xent ignore_index dlogits', xg.grad.cpu(), x64.grad, 1e-5)
    assert torch.count_nonzero(xg.grad[[1, 6, 12]]) == 0
    counter = torch.zeros((), dtype=torch.int32, device=DEV)
    ops.topk_correct(logits.to(DEV), labels.to(DEV), C, counter)           # k = C: every row with a real label is "correct"
    assert int(counter) == B - 3


def test_topk_accuracy_matches_topk():
    from vtx import ops
    g = torch.Generator().manual_seed(3)
    scores = torch.randn(97, 50, generator=g)
    scores[:, 10] = scores[:, 11]                            # ties
    labels = torch.randint(0, 50, (97,), generator=g)
    for k in (1, 5):
        topk = scores.topk(k, dim=-1).indices
        want = int((topk == labels[:, None]).any(-1).sum())
        counter = torch.zeros((), dtype=torch.int32, device=DEV)
        ops.topk_correct(scores.to(DEV), labels.to(DEV), k, counter)
        # torch.topk's tie order is unspecified: rows where the label value is tied at the k-th place may differ
        tied = int(sum(1 for r in range(97) if (scores[r] == scores[r, labels[r]]).sum() > 1))
        assert abs(int(counter) - want) <= tied, (k, int(counter), want)
