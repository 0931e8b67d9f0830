"""Drop-in for the reference ``mixup.py`` (batch-mode Mixup / CutMix + soft targets, reference :16-126)
with the tensor work on libvtx kernels (csrc/head.hip).

The random draws stay on the host in the reference's order -- ``np.random.rand`` (apply?),
``np.random.rand`` (cutmix?), ``np.random.beta`` (lambda), ``np.random.randint`` x2 (box centre) -- so a
seeded run mixes the same clips with the same lambda and box.  On the device, in place and in one pass:
mixup of the clip batch, the CutMix box swap, and the label-smoothed mixed targets; each is
bit-identical to the reference's ATen arithmetic (tests/test_gpu_head.py).
"""
import numpy as np
import torch

from vtx import ops


def one_hot(x, num_classes, on_value=1., off_value=0., device='cuda'):
    x = x.long().view(-1, 1)
    return torch.full((x.size()[0], num_classes), off_value, device=device).scatter_(1, x, on_value)


def mixup_target(target, num_classes, lam=1., smoothing=0.0, device='cuda'):
    if target.is_cuda:
        return ops.mixup_target(target, num_classes, lam, smoothing)
    off_value = smoothing / num_classes                      # host tensors: the reference's formula
    on_value = 1. - smoothing + off_value
    y1 = one_hot(target, num_classes, on_value=on_value, off_value=off_value, device=device)
    y2 = one_hot(target.flip(0), num_classes, on_value=on_value, off_value=off_value, device=device)
    return y1 * lam + y2 * (1. - lam)


def rand_bbox(img_shape, lam, margin=0., count=None):
    """Random square box covering a (1 - lam) fraction of the frame, clipped at the borders."""
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = np.random.randint(0 + margin_y, img_h - margin_y, size=count)
    cx = np.random.randint(0 + margin_x, img_w - margin_x, size=count)
    yl, yh = np.clip(cy - cut_h // 2, 0, img_h), np.clip(cy + cut_h // 2, 0, img_h)
    xl, xh = np.clip(cx - cut_w // 2, 0, img_w), np.clip(cx + cut_w // 2, 0, img_w)
    return yl, yh, xl, xh


def cutmix_bbox_and_lam(img_shape, lam, correct_lam=True, count=None):
    yl, yu, xl, xu = rand_bbox(img_shape, lam, count=count)
    if correct_lam:
        lam = 1. - (yu - yl) * (xu - xl) / float(img_shape[-2] * img_shape[-1])
    return (yl, yu, xl, xu), lam


class Mixup:
    """Batch-mode Mixup / CutMix with label smoothing (reference :59-126; same constructor).
    ``__call__(x [B,T,C,H,W] or [B,C,H,W], target [B]) -> (x mixed in place, soft targets [B, num_classes])``."""

    def __init__(self, mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, mode='batch', correct_lam=True,
                 label_smoothing=0.1, num_classes=1000):
        self.mixup_alpha = mixup_alpha
        self.cutmix_alpha = cutmix_alpha
        self.mix_prob = prob
        self.switch_prob = switch_prob
        self.label_smoothing = label_smoothing
        self.num_classes = num_classes
        self.mode = mode
        self.correct_lam = correct_lam
        self.mixup_enabled = True

    def _params_per_batch(self):
        lam, use_cutmix = 1., False
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
                use_cutmix = np.random.rand() < self.switch_prob
                alpha = self.cutmix_alpha if use_cutmix else self.mixup_alpha
            elif self.mixup_alpha > 0.:
                alpha = self.mixup_alpha
            elif self.cutmix_alpha > 0.:
                use_cutmix, alpha = True, self.cutmix_alpha
            else:
                assert False, 'One of mixup_alpha > 0., cutmix_alpha > 0.'
            lam = float(np.random.beta(alpha, alpha))
        return lam, use_cutmix

    def _mix_batch(self, x):
        lam, use_cutmix = self._params_per_batch()
        if lam == 1.:
            return 1.
        if use_cutmix:
            (yl, yh, xl, xh), lam = cutmix_bbox_and_lam(x.shape, lam, correct_lam=self.correct_lam)
            if x.is_cuda:
                ops.cutmix_batch_(x, yl, yh, xl, xh)
            else:
                x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]
        elif x.is_cuda:
            ops.mixup_batch_(x, lam)
        else:
            x_flipped = x.flip(0).mul_(1. - lam)
            x.mul_(lam).add_(x_flipped)
        return lam

    def __call__(self, x, target):
        assert len(x) % 2 == 0, 'Batch size should be even when using this'
        shape = x.shape
        if x.ndim == 5:
            b, t, c, h, w = shape
            x = x.view(b, t * c, h, w)
        lam = self._mix_batch(x)
        target = mixup_target(target, self.num_classes, lam, self.label_smoothing, x.device)
        return x.view(shape), target
