"""The Lightning entry point of the hot path (drop-in model_trainer.VideoTransformer, reference model_trainer.py:39-310)
driven for one optimisation step the way Lightning drives it: training_step -> backward -> on_after_backward ->
optimizer_step, on a real device, with the fused optimizer, device Mixup and the clip / weight-decay schedule hooks."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _configs(**kw):
    c = dict(objective='supervised', arch='timesformer', pretrain_pth=None, weights_from='imagenet', img_size=32, num_frames=2,
             attention_type='divided_space_time', num_class=10, eval_metrics='finetune', mixup=True, optim_type='sgd', lr=0.05,
             weight_decay=0.05, weight_decay_end=0.01, lr_schedule='cosine', warmup_epochs=2, min_lr=1e-4, clip_grad=0.5,
             layer_decay=1, save_ckpt_freq=1)
    c.update(kw)
    return types.SimpleNamespace(**c)


@pytest.mark.parametrize('mix,optim_type', [(True, 'sgd'), (False, 'adamw')])
def test_one_lightning_step(mix, optim_type):
    import vtx
    import model_trainer as MT
    vtx.set_precision('bf16')
    try:
        torch.manual_seed(0)
        np.random.seed(0)
        trainer = types.SimpleNamespace(max_epochs=10, current_epoch=3, save_checkpoint=lambda p: None)
        m = MT.VideoTransformer(_configs(mixup=mix, optim_type=optim_type), trainer, ckpt_dir='/tmp', do_eval=False, do_test=False)
        with torch.no_grad():
            for blk in m.model.transformer_layers.layers:
                blk.attentions[0].temporal_fc.weight.normal_(0, 0.02)
        m.to(DEV).train()
        (opts, scheds) = m.configure_optimizers()
        opt = opts[0]
        assert type(opt).__name__ == ('FusedSGD' if optim_type == 'sgd' else 'FusedAdamW') and scheds[0] is not None
        assert opt.param_groups[0]['weight_decay'] == 0 and len(opt.param_groups) == 2
        m._optimizers = opt                                          # what Lightning's self.optimizers() returns
        g = torch.Generator().manual_seed(1)
        x = torch.randn(4, 2, 3, 32, 32, generator=g).to(DEV)
        y = torch.randint(0, 10, (4,), generator=g).to(DEV)
        before = {k: v.detach().clone() for k, v in m.named_parameters()}
        out = m.training_step((x, y), 0)
        loss = out['loss']
        assert loss.ndim == 0 and torch.isfinite(loss)
        loss.backward()
        m.on_after_backward()
        # the logged statistic is the reference's: the 2-norm of the per-parameter gradient norms
        norms = torch.stack([p.grad.double().norm() for p in m.parameters() if p.grad is not None])
        want = norms.norm().item()
        got = float(m.logged['grad_norm'])
        assert abs(got - want) <= 1e-4 * want, (got, want)
        # cosine weight-decay schedule, decayed group only (model_trainer.py:144-151)
        import math
        wd = 0.01 - (0.01 - 0.05) * (math.cos(math.pi * 3 / 10) + 1) / 2
        assert abs(opt.param_groups[1]['weight_decay'] - wd) < 1e-12 and opt.param_groups[0]['weight_decay'] == 0
        m.optimizer_step(3, 0, opt, 0, None, False, True, False)
        assert m.iteration == 1
        moved = sum(1 for k, v in m.named_parameters() if v.grad is not None and not torch.equal(v.detach(), before[k]))
        assert moved == sum(1 for v in m.parameters() if v.grad is not None) > 200
        assert 0.0 <= float(m.logged['top5_acc']) <= 1.0
        # epoch-wise warm-up of the reference's cosine schedule: lr = base * (epoch + 1) / warmup_epochs
        assert abs(opt.param_groups[0]['lr'] - 0.025) < 1e-12
        scheds[0].step()
        assert abs(opt.param_groups[0]['lr'] - 0.05) < 1e-12
    finally:
        vtx.set_precision('auto')
