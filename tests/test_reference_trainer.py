"""Boundary b1 in the dev container: the REFERENCE's own Lightning module (model_trainer.py, unmodified, with
pytorch_lightning / torchmetrics / timm / torchvision stubbed exactly as absent packages) builds its model, head,
Mixup, loss and optimizer ON TOP OF the drop-in transformer.py / video_transformer.py / optimizer.py / mixup.py /
utils.py of this package -- i.e. the reference's entry point accepts the drop-in files.  (Its training_step needs a
GPU; the same step is driven on the device through the drop-in model_trainer.py in tests/test_gpu_trainer.py.)"""
import importlib.util
import os
import sys
import types

import pytest
import torch.nn as nn

from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason='needs /root/reference (dev container)')


def _load_reference_trainer():
    stubs = {}

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        stubs[name] = m
        return m
    stub('pytorch_lightning', LightningModule=nn.Module)
    stub('torchvision')

    class Accuracy:
        def __init__(self, top_k=1):
            self.top_k = top_k
    stub('torchmetrics', Accuracy=Accuracy)
    stub('timm')
    stub('timm.loss', SoftTargetCrossEntropy=type('SoftTargetCrossEntropy', (nn.Module,), {}))
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location('ref_model_trainer', os.path.join(ref_loader.REF_DIR, 'model_trainer.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)       # its `from transformer import ...` etc. resolve to the drop-in package on sys.path
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


@pytest.mark.parametrize('arch', ['timesformer', 'vivit'])
def test_reference_lightning_module_builds_on_the_drop_in_modules(arch):
    import transformer
    import video_transformer
    MT = _load_reference_trainer()
    assert MT.TimeSformer is video_transformer.TimeSformer and MT.ClassificationHead is transformer.ClassificationHead
    cfg = types.SimpleNamespace(objective='supervised', arch=arch, pretrain_pth=None, weights_from='imagenet', img_size=32,
                                num_frames=4, attention_type='divided_space_time' if arch == 'timesformer' else 'fact_encoder',
                                num_class=174, eval_metrics='finetune', mixup=True, optim_type='adamw', lr=1e-3, weight_decay=0.05,
                                weight_decay_end=0.05, lr_schedule='cosine', warmup_epochs=1, min_lr=1e-5, clip_grad=1.0,
                                layer_decay=1)
    trainer = types.SimpleNamespace(max_epochs=5, current_epoch=0)
    m = MT.VideoTransformer(cfg, trainer, ckpt_dir='/tmp', do_eval=True, do_test=False)
    assert type(m.model) is getattr(video_transformer, 'TimeSformer' if arch == 'timesformer' else 'ViViT')
    assert m.cls_head.cls_head.out_features == 174
    opts, scheds = m.configure_optimizers()
    assert type(opts[0]).__name__ == 'FusedAdamW' and len(opts[0].param_groups) == 2
    assert scheds[0] is not None
    assert m.no_weight_decay_keywords() == m.model.no_weight_decay_keywords()


def test_drop_in_mixup_plans_a_batch_like_the_reference_draw_for_draw():
    """The host half of mixup.Mixup (``draw``) against the reference's class under the same numpy seed: same lambda, same
    CutMix box, same generator state afterwards (reference mixup.py:74-88,:105-109).  The device half is proven equal to
    the reference's tensor formulas in tests/test_gpu_head.py; this package has no CPU tensor path."""
    import numpy as np
    import mixup as ours
    spec = importlib.util.spec_from_file_location('ref_mixup', os.path.join(ref_loader.REF_DIR, 'mixup.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    shape = (4, 6, 16, 16)
    kinds = set()
    for kw in (dict(), dict(cutmix_alpha=0.), dict(mixup_alpha=0.), dict(prob=0.5), dict(correct_lam=False)):
        for seed in range(8):
            np.random.seed(seed)
            rm = ref.Mixup(num_classes=12, **kw)
            lam_r, cut = rm._params_per_batch()
            box_r = None
            if lam_r != 1. and cut:
                box_r, lam_r = ref.cutmix_bbox_and_lam(shape, lam_r, correct_lam=rm.correct_lam)
            state_r = np.random.get_state()[1].copy()
            np.random.seed(seed)
            lam_o, box_o = ours.Mixup(num_classes=12, **kw).draw(shape)
            assert np.array_equal(np.random.get_state()[1], state_r), (kw, seed)
            assert float(lam_o) == float(lam_r), (kw, seed)
            assert (box_o is None) == (box_r is None) and (box_o is None or box_o == tuple(int(v) for v in box_r)), (kw, seed)
            kinds.add(box_o is not None)
    assert kinds == {True, False}, 'both kinds of plan must have been exercised'
