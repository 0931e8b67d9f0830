"""Drop-in for the reference ``model_trainer.py``: the Lightning module that drives the hot path
(reference :39-310), same constructor, hooks and ``configs`` fields, on top of the drop-in models.

What changes underneath (nothing in the interface):
  * models / head: ``video_transformer.py`` / ``transformer.py`` of this package (HIP kernels);
  * ``configure_optimizers``: ``optimizer.build_optimizer`` -> multi-tensor ``vtx.optim`` optimizers; the cosine
    schedule and ``MultiStepLR`` are the reference's (torch schedulers work on any ``Optimizer``);
  * ``clip_gradients``: per-parameter norms in two kernel launches instead of ~250 ``torch.norm`` calls; the
    clip itself is applied inside the fused optimizer step (the stored ``.grad`` is not rewritten);
  * Mixup / CutMix, soft-target cross-entropy, top-k accuracy: ``mixup.py`` / ``vtx`` kernels on the device
    (the reference imports timm's loss and torchmetrics, neither of which this path needs).
pytorch_lightning is optional: without it the class derives from a minimal stand-in with the members the
hooks use (``log``, ``print``, ``optimizers()``), so the module can be driven by any loop -- that is how
tests/test_gpu_trainer.py runs ``training_step`` / ``on_after_backward`` / ``optimizer_step``.
"""
import math
import os.path as osp
import time

import torch
import torch.nn as nn
import torch.optim as optim

import utils  # noqa: F401  (kept: the reference module exposes it)
from mixup import Mixup
from optimizer import build_optimizer
from transformer import ClassificationHead
from video_transformer import TimeSformer, ViViT, MaskFeat
from vtx import functions as F_
from vtx import ops

try:
    import pytorch_lightning as pl
except ImportError:                                      # minimal stand-in (see module docstring)
    class _LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.logged = {}
            self._optimizers = None

        def log(self, name, value, **kwargs):
            self.logged[name] = value

        def print(self, *args, **kwargs):
            print(*args, **kwargs)

        def optimizers(self):
            class _Wrap:
                def __init__(self, o):
                    self.optimizer = o
            if self._optimizers is None:
                self._optimizers = self.configure_optimizers()[0][0]
            return _Wrap(self._optimizers)

    class pl:                                            # noqa: N801
        LightningModule = _LightningModule


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, base_lr, objective, min_lr=5e-5,
                                    last_epoch=-1):
    """Epoch-wise linear warm-up to base_lr, then half a cosine down to 0 ('mim') or to min_lr (reference :20-37)."""
    def lr_lambda(epoch):
        epoch += 1
        if epoch <= num_warmup_steps:
            return float(epoch) / float(max(1, num_warmup_steps))
        progress = min(float(epoch - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps)), 1)
        factor = 0.5 * (1. + math.cos(math.pi * progress))
        if objective == 'mim':
            return factor
        return factor * (1 - min_lr / base_lr) + min_lr / base_lr
    return optim.lr_scheduler.LambdaLR(optimizer, lr_lambda, last_epoch)


class Accuracy:
    """Top-k accuracy accumulated on the device (the torchmetrics.Accuracy calls of the reference:
    ``acc(probs, labels)`` returns the batch accuracy and accumulates; ``compute()``; ``reset()``)."""

    def __init__(self, top_k=1):
        self.top_k = top_k
        self.correct = None
        self.total = 0

    def __call__(self, scores, labels):
        if self.correct is None:
            self.correct = torch.zeros((), dtype=torch.int32, device=scores.device)
        before = self.correct.clone()
        ops.topk_correct(scores, labels, self.top_k, self.correct)
        self.total += scores.shape[0]
        return (self.correct - before).float() / scores.shape[0]

    def compute(self):
        """Accuracy over everything seen since reset() -- over ALL data-parallel ranks when a process group is up, as
        torchmetrics synchronises its states in compute() (the reference selects its best checkpoint from this value)."""
        if self.correct is None:
            return torch.zeros(())
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            st = torch.stack([self.correct.float(), torch.tensor(float(self.total), device=self.correct.device)])
            from vtx import dp
            dp.all_reduce_sum(st)                       # (staged through host memory when the backend is not RCCL)
            return st[0] / st[1].clamp(min=1.0)
        return self.correct.float() / max(self.total, 1)

    def reset(self):
        self.correct, self.total = None, 0


class SoftTargetCrossEntropy(nn.Module):
    """mean_b sum_c -t log_softmax(x) (timm.loss.SoftTargetCrossEntropy, reference :87-88)."""

    def forward(self, x, target):
        return F_.SoftmaxXentFn.apply(x, target)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() with default arguments on int64 labels (reference :91)."""

    def forward(self, x, target):
        return F_.SoftmaxXentFn.apply(x, target)


class VideoTransformer(pl.LightningModule):
    def __init__(self, configs, trainer, ckpt_dir, do_eval, do_test, n_crops=3):
        super().__init__()
        self.configs = configs
        self.trainer = trainer
        if configs.objective == 'mim':
            self.model = MaskFeat(pool_q_stride_size=[[1, 1, 2, 2], [3, 1, 2, 2]], feature_dim=2 * 2 * 2 * 3 * 9)
        else:
            if configs.arch == 'vivit':
                self.model = ViViT(pretrain_pth=configs.pretrain_pth, weights_from=configs.weights_from,
                                   img_size=configs.img_size, num_frames=configs.num_frames,
                                   attention_type=configs.attention_type)
            elif configs.arch == 'timesformer':
                self.model = TimeSformer(pretrain_pth=configs.pretrain_pth, weights_from=configs.weights_from,
                                         img_size=configs.img_size, num_frames=configs.num_frames,
                                         attention_type=configs.attention_type)
            else:                                        # 'mvit': the MaskFeat backbone fine-tuned, decoder frozen
                self.model = MaskFeat(pool_q_stride_size=[[1, 1, 2, 2], [3, 1, 2, 2]], feature_dim=2 * 2 * 2 * 3 * 9,
                                      pretrain_pth=configs.pretrain_pth, img_size=configs.img_size,
                                      num_frames=configs.num_frames)
                for param in self.model.decoder_pred.parameters():
                    param.requires_grad = False
            self.cls_head = ClassificationHead(configs.num_class, self.model.embed_dims, eval_metrics=configs.eval_metrics)
            self.max_top1_acc = 0
            self.train_top1_acc = Accuracy()
            self.train_top5_acc = Accuracy(top_k=5)
            if configs.mixup:
                self.mixup_fn = Mixup(num_classes=configs.num_class)
                self.loss_fn = SoftTargetCrossEntropy()
            else:
                self.loss_fn = CrossEntropyLoss()
        self.iteration = 0
        self.data_start = 0
        self.ckpt_dir = ckpt_dir
        self.do_eval = do_eval
        self.do_test = do_test
        if do_eval:
            self.val_top1_acc = Accuracy()
            self.val_top5_acc = Accuracy(top_k=5)
        if do_test:
            self.n_crops = n_crops
            self.test_top1_acc = Accuracy()
            self.test_top5_acc = Accuracy(top_k=5)

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'pos_embed', 'cls_token', 'mask_token'}

    def _optimised_module(self):
        if self.configs.objective == 'supervised' and self.configs.eval_metrics == 'linear_prob':
            return self.cls_head.module if hasattr(self.cls_head, 'module') else self.cls_head
        return self.module if hasattr(self, 'module') else self

    def configure_optimizers(self):
        is_pretrain = self.configs.objective != 'supervised'
        target = self._optimised_module() if not is_pretrain and self.configs.eval_metrics == 'linear_prob' else self
        optimizer = build_optimizer(self.configs, target, is_pretrain=is_pretrain)
        lr_scheduler = None
        if self.configs.lr_schedule == 'multistep':
            lr_scheduler = optim.lr_scheduler.MultiStepLR(optimizer, milestones=[5, 11], gamma=0.1)
        elif self.configs.lr_schedule == 'cosine':
            lr_scheduler = get_cosine_schedule_with_warmup(
                optimizer, num_warmup_steps=self.configs.warmup_epochs, num_training_steps=self.trainer.max_epochs,
                base_lr=self.configs.lr, min_lr=self.configs.min_lr, objective=self.configs.objective)
        return [optimizer], [lr_scheduler]

    def parse_batch(self, batch, train):
        if self.configs.objective == 'mim':
            inputs, labels, mask, cube_marker = batch
            return inputs, labels, mask, cube_marker
        inputs, labels = batch
        if self.configs.mixup and train:
            inputs, labels = self.mixup_fn(inputs, labels)
        return inputs, labels

    # ---- epoch schedules -----------------------------------------------------------------------
    def _get_momentum(self, base_value, final_value):
        phase = math.cos(math.pi * self.trainer.current_epoch / self.trainer.max_epochs)
        return final_value - (final_value - base_value) * (phase + 1) / 2

    def _weight_decay_update(self):
        groups = self.optimizers().optimizer.param_groups
        if len(groups) > 1:                              # only the decayed group (index 1) follows the schedule
            groups[1]['weight_decay'] = self._get_momentum(self.configs.weight_decay, self.configs.weight_decay_end)

    def clip_gradients(self, clip_grad, norm_type=2):
        """Norm of the per-parameter gradient norms (what the reference logs); with ``clip_grad`` every parameter's
        gradient is scaled by min(1, clip_grad / (its norm + 1e-6)) -- here inside the fused optimizer step."""
        if norm_type != 2:
            raise NotImplementedError('vtx: clip_gradients supports the 2-norm (the only one the reference uses)')
        opt = self.optimizers().optimizer
        opt.clip_grad = clip_grad if clip_grad else None
        return opt.grad_norm()

    def log_step_state(self, data_time, top1_acc=0, top5_acc=0):
        self.log('time', float(f'{time.perf_counter() - self.data_start:.3f}'), prog_bar=True)
        self.log('data_time', data_time, prog_bar=True)
        if self.configs.objective == 'supervised':
            self.log('top1_acc', top1_acc, on_step=True, on_epoch=False, prog_bar=True)
            self.log('top5_acc', top5_acc, on_step=True, on_epoch=False, prog_bar=True)

    def get_progress_bar_dict(self):
        items = super().get_progress_bar_dict()
        items.pop('v_num', None)
        return items

    # ---- trainer pipeline ----------------------------------------------------------------------
    def _features(self, inputs):
        if self.configs.arch == 'mvit':
            return self.model.forward_features(inputs)[:, 0]
        return self.model(inputs)

    def training_step(self, batch, batch_idx):
        data_time = float(f'{time.perf_counter() - self.data_start:.3f}')
        if self.configs.objective == 'mim':
            inputs, labels, mask, cube_marker = self.parse_batch(batch, train=True)
            preds, loss = self.model(inputs, labels, mask, cube_marker)
            self.log_step_state(data_time)
            return {'loss': loss, 'data_time': data_time}
        inputs, labels = self.parse_batch(batch, train=True)
        if self.configs.eval_metrics == 'linear_prob':
            with torch.no_grad():
                self.model.eval()
                preds = self.model(inputs)
        else:
            preds = self._features(inputs)
        preds = self.cls_head(preds)
        loss = self.loss_fn(preds, labels)
        hard = labels.argmax(-1) if self.configs.mixup else labels
        # softmax is monotone per row: the top-k of the logits is the top-k of preds.softmax(-1)
        top1_acc = self.train_top1_acc(preds.detach(), hard)
        top5_acc = self.train_top5_acc(preds.detach(), hard)
        self.log_step_state(data_time, top1_acc, top5_acc)
        return {'loss': loss, 'data_time': data_time}

    def on_after_backward(self):
        param_norms = self.clip_gradients(self.configs.clip_grad)
        self._weight_decay_update()
        self.log('lr', self.optimizers().optimizer.param_groups[0]['lr'], on_step=True, on_epoch=False, prog_bar=True)
        self.log('grad_norm', param_norms, on_step=True, on_epoch=False, prog_bar=True)

    def optimizer_step(self, epoch, batch_idx, optimizer, optimizer_idx, optimizer_closure, on_tpu, using_native_amp,
                       using_lbfgs):
        optimizer.step(closure=optimizer_closure)
        self.data_start = time.perf_counter()
        self.iteration += 1

    def _stamp(self):
        return time.strftime('%Y-%m-%d %H:%M:%S', time.localtime())

    def training_epoch_end(self, outputs):
        stamp = self._stamp()
        if self.configs.objective == 'supervised':
            self.print(f'{stamp} - Evaluating mean ', f'top1_acc:{self.train_top1_acc.compute():.3f},',
                       f'top5_acc:{self.train_top5_acc.compute():.3f} of current training epoch')
            self.train_top1_acc.reset()
            self.train_top5_acc.reset()
        self.trainer.save_checkpoint(osp.join(self.ckpt_dir, 'last_checkpoint.pth'))
        if self.configs.objective != 'supervised' and (self.trainer.current_epoch + 1) % self.configs.save_ckpt_freq == 0:
            self.trainer.save_checkpoint(osp.join(self.ckpt_dir, f'{stamp}_ep_{self.trainer.current_epoch}.pth'))

    def validation_step(self, batch, batch_indx):
        if not self.do_eval:
            return
        inputs, labels = self.parse_batch(batch, train=False)
        if self.configs.eval_metrics == 'linear_prob':
            with torch.no_grad():
                preds = self.model(inputs)
        else:
            preds = self._features(inputs)
        preds = self.cls_head(preds)
        self.val_top1_acc(preds, labels)
        self.val_top5_acc(preds, labels)
        self.data_start = time.perf_counter()

    def validation_epoch_end(self, outputs):
        if not self.do_eval:
            return
        top1, top5 = self.val_top1_acc.compute(), self.val_top5_acc.compute()
        stamp = self._stamp()
        self.print(f'{stamp} - Evaluating mean ', f'top1_acc:{top1:.3f}, ', f'top5_acc:{top5:.3f} of current validation epoch')
        self.val_top1_acc.reset()
        self.val_top5_acc.reset()
        if top1 > self.max_top1_acc:                     # best checkpoint so far
            self.trainer.save_checkpoint(osp.join(self.ckpt_dir, f'{stamp}_ep_{self.trainer.current_epoch}_top1_acc_{top1:.3f}.pth'))
            self.max_top1_acc = top1

    def test_step(self, batch, batch_idx):
        if not self.do_test:
            return
        inputs, labels = self.parse_batch(batch, train=False)
        preds = self.cls_head(self.model(inputs))
        preds = preds.view(-1, self.n_crops, self.configs.num_class).mean(1)     # average the crops of a clip
        self.test_top1_acc(preds, labels)
        self.test_top5_acc(preds, labels)
        self.data_start = time.perf_counter()

    def test_epoch_end(self, outputs):
        if not self.do_test:
            return
        self.print(f'{self._stamp()} - Evaluating mean ', f'top1_acc:{self.test_top1_acc.compute():.3f}, ',
                   f'top5_acc:{self.test_top5_acc.compute():.3f} of current test epoch')
        self.test_top1_acc.reset()
        self.test_top5_acc.reset()
