"""Drop-in for the reference ``optimizer.py`` (SimMIM-style parameter groups, reference :14-166):
same entry points and grouping rules, returning the multi-tensor ``vtx.optim`` optimizers (one
kernel launch per step instead of a Python loop / foreach over ~250 parameters).

    build_optimizer(hparams, model, is_pretrain)
        hparams: .optim_type ('sgd' | 'adamw'), .lr, .weight_decay (+ .arch, .layer_decay for fine-tuning)

Grouping (reference :42-62, :113-160): a parameter is NOT decayed when it is 1-D, a ``.bias``, named in
``model.no_weight_decay()`` or contains one of ``model.no_weight_decay_keywords()``
({'pos_embed', 'cls_token', 'mask_token'} for every model here).  The no-decay group comes first, so
``param_groups[1]`` is the decayed one the trainer's cosine weight-decay schedule rewrites
(model_trainer.py:147-151).  MViT fine-tuning adds layer-wise lr decay: one pair of groups per
layer id (patch embed / positional encoding / mask token 0, block i -> i + 1, head last).
"""
from functools import partial

from utils import print_on_rank_zero
from vtx.optim import FusedAdamW, FusedSGD


def check_keywords_in_name(name, keywords=()):
    return any(k in name for k in keywords)


def _skips(model):
    skip = model.no_weight_decay() if hasattr(model, 'no_weight_decay') else {}
    kw = model.no_weight_decay_keywords() if hasattr(model, 'no_weight_decay_keywords') else {}
    return skip, kw


def _no_decay(name, param, skip_list, skip_keywords):
    return len(param.shape) == 1 or name.endswith('.bias') or name in skip_list or check_keywords_in_name(name, skip_keywords)


def _make(hparams, groups):
    kind = hparams.optim_type.lower()
    if kind == 'sgd':
        return FusedSGD(groups, momentum=0.9, nesterov=True, lr=hparams.lr, weight_decay=hparams.weight_decay)
    if kind == 'adamw':
        return FusedAdamW(groups, betas=(0.9, 0.999), lr=hparams.lr, weight_decay=hparams.weight_decay)
    return None                                     # the reference returns None for an unknown optim_type too


def get_pretrain_param_groups(model, skip_list=(), skip_keywords=()):
    plain, decayed = ([], []), ([], [])
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        params, names = plain if _no_decay(name, param, skip_list, skip_keywords) else decayed
        params.append(param)
        names.append(name)
    print_on_rank_zero(f'params_no_decay_name: {plain[1]} \n params_decay_name: {decayed[1]}')
    return [{'params': plain[0], 'weight_decay': 0.}, {'params': decayed[0]}]


def build_pretrain_optimizer(hparams, model):
    skip, kw = _skips(model)
    return _make(hparams, get_pretrain_param_groups(model, skip, kw))


def get_mvit_layer(name, num_layers):
    """Layer id of an MViT parameter for layer-wise lr decay (reference :100-111)."""
    name = name.replace('mvit.', '').replace('model.', '')
    if name in ('mask_token') or name.startswith('patch_embed') or name.startswith('cls_positional_encoding'):
        return 0
    if name.startswith('blocks'):
        return int(name.split('.')[1]) + 1
    return num_layers - 1


def get_finetune_param_groups(model, lr, weight_decay, get_layer_func, scales, skip_list=(), skip_keywords=()):
    groups = {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        plain = _no_decay(name, param, skip_list, skip_keywords)
        layer_id = get_layer_func(name) if get_layer_func is not None else None
        gname = ('no_decay' if plain else 'decay') if layer_id is None else 'layer_%d_%s' % (layer_id, 'no_decay' if plain else 'decay')
        if gname not in groups:
            scale = scales[layer_id] if scales is not None else 1.
            groups[gname] = {'group_name': gname, 'weight_decay': 0. if plain else weight_decay, 'params': [],
                             'lr': lr * scale, 'lr_scale': scale}
        groups[gname]['params'].append(param)
    return list(groups.values())


def build_finetune_optimizer(hparams, model):
    if hparams.arch != 'mvit':
        return build_pretrain_optimizer(hparams, model)
    layer_fn = scales = None
    if hparams.layer_decay != 1:
        depth = 16
        layer_fn = partial(get_mvit_layer, num_layers=depth + 2)
        scales = [hparams.layer_decay ** i for i in reversed(range(depth + 2))]
    skip, kw = _skips(model)
    return _make(hparams, get_finetune_param_groups(model, hparams.lr, hparams.weight_decay, layer_fn, scales, skip, kw))


def build_optimizer(hparams, model, is_pretrain):
    return build_pretrain_optimizer(hparams, model) if is_pretrain else build_finetune_optimizer(hparams, model)
