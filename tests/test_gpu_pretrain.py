"""The drop-in ``model_pretrain.single_run()`` (reference model_pretrain.py:154-230) on one GPU: the reference's flags in,
``model_trainer.VideoTransformer`` built, parameters broadcast, gradients bucketed and all-reduced through a REAL RCCL group
(1 rank, VTX_FORCE_DP=1: the collectives are issued and waited for), the Lightning hook order per step, epoch-wise LR
schedule, checkpoint + resume.  N > 1 ranks take the same code path with WORLD_SIZE > 1 (tests/test_dp_gloo.py covers
the cross-rank reduction on CPU; the 8-GPU run is the driver's)."""
import math
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _argv(root, **kw):
    a = {'-epoch': 2, '-batch_size': 4, '-root_dir': root, '-num_class': 10, '-num_frames': 2, '-frame_interval': 4,
         '-train_data_path': 'synthetic', '-lr': 0.64, '-objective': 'supervised', '-img_size': 32, '-optim_type': 'sgd',
         '-synthetic_steps': 2, '-gpus': 0, '-clip_grad': 0.5, '-warmup_epochs': 1, '-log_interval': 1, '-mixup': 1}
    a.update(kw)
    out = []
    for k, v in a.items():
        out += [k] if v is True else [k, str(v)]
    return out


def test_single_run_supervised_two_epochs_then_resume(tmp_path, monkeypatch):
    import model_pretrain as MP
    monkeypatch.setenv('VTX_FORCE_DP', '1')
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', _port())
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    tr = MP.single_run(_argv(str(tmp_path)))
    assert tr.global_step == 4 and tr.current_epoch == 2 and torch.isfinite(tr.last_loss)
    assert not torch.distributed.is_initialized(), 'single_run tears its process group down'
    assert type(tr.optimizer).__name__ == 'FusedSGD'
    # linear LR scaling (reference :158-164): 0.64 * (4 clips x 1 GPU) / 256 = 0.01; cosine schedule with 1 warm-up epoch,
    # stepped per epoch (model_trainer.get_cosine_schedule_with_warmup)
    assert abs(tr.args.lr - 0.01) < 1e-12
    ckpt = os.path.join(str(tmp_path), 'results', MP.experiment_tag(tr.args), 'ckpt', 'last_checkpoint.pth')
    state = torch.load(ckpt, map_location='cpu')
    assert state['epoch'] == 2 and state['global_step'] == 4 and len(state['state_dict']) > 240    # Lightning: current_epoch + 1
    moved = tr.model.model.transformer_layers.layers[3].ffns[0].layers[1].weight.detach().clone()
    # resume: one more epoch from the checkpoint (the tag is computed from the scaled lr, so pass the path)
    monkeypatch.setenv('MASTER_PORT', _port())
    tr2 = MP.single_run(_argv(str(tmp_path), **{'-epoch': 3, '-resume_from_checkpoint': ckpt}))
    assert tr2.current_epoch == 3 and tr2.global_step == 6, (tr2.current_epoch, tr2.global_step)
    w2 = tr2.model.model.transformer_layers.layers[3].ffns[0].layers[1].weight.detach()
    assert not torch.equal(w2.cpu(), moved.cpu()) and torch.isfinite(w2).all()
    # the resumed scheduler continued (epoch 2 of 3 on the cosine part), it did not restart its warm-up
    lr_now = tr2.optimizer.param_groups[0]['lr']
    assert 0 < lr_now < 0.01 and not math.isclose(lr_now, 0.01)


def test_single_run_mim_one_step(tmp_path, monkeypatch):
    """objective mim: MaskFeat / MViT-B on 16x224^2 synthetic clips with on-device HOG targets, AdamW with layer decay."""
    import model_pretrain as MP
    monkeypatch.setenv('VTX_FORCE_DP', '1')
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', _port())
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    tr = MP.single_run(_argv(str(tmp_path), **{'-epoch': 1, '-batch_size': 2, '-objective': 'mim', '-arch': 'mvit', '-num_frames': 16,
                                               '-img_size': 224, '-optim_type': 'adamw', '-lr': 0.1, '-synthetic_steps': 2,
                                               '-save_ckpt_freq': 1, '-clip_grad': 0.02}))
    assert tr.global_step == 2 and torch.isfinite(tr.last_loss) and float(tr.last_loss) > 0
    assert type(tr.optimizer).__name__ == 'FusedAdamW'
    assert type(tr.model.model).__name__ == 'MaskFeat'
