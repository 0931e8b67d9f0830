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


def test_mixup_class_gpu_equals_cpu_draw_for_draw():
    """The drop-in Mixup on CUDA tensors (kernels) and on CPU tensors (the reference's formulas) consume the numpy
    generator identically and produce identical clips and targets."""
    import mixup
    fn = mixup.Mixup(num_classes=12)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(4, 2, 3, 16, 16, generator=g)
    y0 = torch.tensor([1, 5, 5, 11])
    for seed in range(6):                                    # both mixup and cutmix branches occur
        np.random.seed(seed)
        xc, yc = fn(x0.clone(), y0)
        np.random.seed(seed)
        xg, yg = fn(x0.clone().to(DEV), y0.to(DEV))
        assert torch.equal(xg.cpu(), xc) and torch.equal(yg.cpu(), yc), seed
        assert xg.shape == x0.shape and yg.shape == (4, 12)


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
