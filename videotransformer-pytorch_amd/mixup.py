"""Drop-in for the reference ``mixup.py`` (batch-mode Mixup / CutMix + soft targets, reference :16-126): the same
public names and constructor, the tensor work on libvtx kernels (csrc/head.hip), device tensors only.

Split in two halves:
  * the HOST half decides what happens to a batch -- ``Mixup.draw`` -- and consumes ``np.random`` exactly as the reference
    does (``rand`` apply?, ``rand`` cutmix?, ``beta`` lambda, then for CutMix ``randint`` row, ``randint`` column), so a
    seeded run mixes the same clips with the same lambda and box (tests/test_reference_trainer.py compares the plan
    and the generator state with the reference's class);
  * the DEVICE half applies the plan in place, one pass each: ``vtx_mixup_batch``, ``vtx_cutmix_batch``,
    ``vtx_mixup_target`` -- bit-identical to the reference's ATen arithmetic (tests/test_gpu_head.py).
CPU tensors raise (vtx.ops.need_cuda): this package has no host fallback.
"""
import numpy as np
import torch

from vtx import ops


def one_hot(x, num_classes, on_value=1., off_value=0., device='cuda'):
    """[B, num_classes] fp32 rows holding ``on_value`` at the label and ``off_value`` elsewhere (reference :16-18)."""
    ops.need_cuda(x)
    labels = x.long().contiguous().view(-1)
    out = labels.new_empty((labels.numel(), num_classes), dtype=torch.float32)
    ops.call('vtx_mixup_target', ops.ptr(labels), labels.numel(), int(num_classes), float(np.float32(on_value)),
             float(np.float32(off_value)), 1.0, 0.0, ops.ptr(out), ops.stream())
    return out


def mixup_target(target, num_classes, lam=1., smoothing=0.0, device='cuda'):
    """Label-smoothed one-hot rows of the batch mixed with those of the flipped batch (reference :20-25)."""
    return ops.mixup_target(target, num_classes, lam, smoothing)


def rand_bbox(img_shape, lam, margin=0., count=None):
    """Box of side sqrt(1 - lam) x the frame around a random centre, clipped to the frame -> (yl, yh, xl, xh).
    Two ``np.random.randint`` draws: the centre row, then the centre column (reference :28-48)."""
    h, w = img_shape[-2], img_shape[-1]
    side = np.sqrt(1 - lam)
    bh, bw = int(h * side), int(w * side)
    my, mx = int(margin * bh), int(margin * bw)
    cy = np.random.randint(my, h - my, size=count)
    cx = np.random.randint(mx, w - mx, size=count)
    rows = np.clip(np.stack([cy - bh // 2, cy + bh // 2]), 0, h)
    cols = np.clip(np.stack([cx - bw // 2, cx + bw // 2]), 0, w)
    return rows[0], rows[1], cols[0], cols[1]


def cutmix_bbox_and_lam(img_shape, lam, correct_lam=True, count=None):
    """The box and the lambda that matches its clipped area (reference :51-56)."""
    box = rand_bbox(img_shape, lam, count=count)
    if correct_lam:
        yl, yh, xl, xh = box
        lam = 1. - (yh - yl) * (xh - xl) / float(img_shape[-2] * img_shape[-1])
    return box, lam


class Mixup:
    """Batch-mode Mixup / CutMix with label smoothing (reference :59-126; same constructor).
    ``__call__(x [B,T,C,H,W] or [B,C,H,W], target [B]) -> (x mixed in place, soft targets [B, num_classes])``."""

    def __init__(self, mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, mode='batch', correct_lam=True,
                 label_smoothing=0.1, num_classes=1000):
        if mode != 'batch':
            raise NotImplementedError("Mixup: only mode='batch' (the one the reference trainer uses, model_trainer.py:66-70)")
        self.mixup_alpha, self.cutmix_alpha = mixup_alpha, cutmix_alpha
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.correct_lam, self.label_smoothing, self.num_classes = correct_lam, label_smoothing, num_classes
        self.mode = mode
        self.mixup_enabled = True

    def draw(self, frame_shape):
        """The plan for one batch: ``(lam, box)`` with box = None for Mixup (or nothing, lam = 1) and (yl, yh, xl, xh)
        for CutMix (lam then matches the box).  Consumes ``np.random`` as the reference's ``_params_per_batch`` +
        ``cutmix_bbox_and_lam`` do (:74-88,:105-109)."""
        if not (self.mixup_enabled and np.random.rand() < self.mix_prob):
            return 1., None
        alphas = {False: self.mixup_alpha, True: self.cutmix_alpha}
        if alphas[False] <= 0. and alphas[True] <= 0.:
            raise AssertionError('One of mixup_alpha > 0., cutmix_alpha > 0.')
        if alphas[False] > 0. and alphas[True] > 0.:
            cut = bool(np.random.rand() < self.switch_prob)
        else:
            cut = alphas[True] > 0.
        lam = float(np.random.beta(alphas[cut], alphas[cut]))
        if not cut or lam == 1.:
            return lam, None
        box, lam = cutmix_bbox_and_lam(frame_shape, lam, correct_lam=self.correct_lam)
        return lam, tuple(int(v) for v in box)

    def __call__(self, x, target):
        ops.need_cuda(x, target)
        if len(x) % 2:
            raise AssertionError('Batch size should be even when using this')
        planes = x.view(x.shape[0], -1, x.shape[-2], x.shape[-1]) if x.ndim == 5 else x        # a view: mixed in place
        lam, box = self.draw(planes.shape)
        if box is not None:
            ops.cutmix_batch_(planes, *box)
        elif lam != 1.:
            ops.mixup_batch_(planes, lam)
        return x, ops.mixup_target(target, self.num_classes, lam, self.label_smoothing)
