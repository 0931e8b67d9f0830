"""Helpers of the reference utils.py under the same names (rank helpers :11-34, timing :36-40,
parameter listing / grouping :42-65, denormalisation and clip plotting :68-127), without its
top-level pytorch_lightning / matplotlib imports: ``import utils`` must work wherever the drop-in
modules do.
"""
import os
import os.path as osp
import time

import numpy as np
import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def print_on_rank_zero(content):
    if is_main_process():
        print(content)


def timeit_wrapper(func, *args, **kwargs):
    """(return value, seconds rounded to 4 decimals)."""
    t0 = time.perf_counter()
    ret = func(*args, **kwargs)
    return ret, float(f'{time.perf_counter() - t0:.4f}')


def show_trainable_params(named_parameters):
    for name, param in named_parameters:
        print(name, param.size())


def build_param_groups(model):
    """[no-decay group (weight_decay 0), decay group]: 1-element tensors and biases are not decayed."""
    groups = ([], []), ([], [])
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        params, names = groups[0] if (len(param) == 1 or name.endswith('.bias')) else groups[1]
        params.append(param)
        names.append(name)
    print_on_rank_zero(f'params_no_decay_name: {groups[0][1]} \n params_decay_name: {groups[1][1]}')
    return [{'params': groups[0][0], 'weight_decay': 0}, {'params': groups[1][0]}]


def denormalize(data, mean, std):
    """x * std + mean over the last (channel) axis of an [..., C] image / video tensor."""
    def as_row(v):
        if isinstance(v, tuple):
            v = torch.tensor(np.array(v, dtype=float), device=data.device, dtype=data.dtype)
        return v[None, :] if v.shape else v
    shape = data.shape
    return (data.contiguous().view(-1, shape[-1]) * as_row(std) + as_row(mean)).view(shape)
