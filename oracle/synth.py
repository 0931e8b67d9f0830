"""Deterministic synthetic weights / inputs shared by the golden generator, the
oracle tests and the GPU parity tests (TEST INFRASTRUCTURE ONLY).

Weights are NOT the modules' default init: ``temporal_fc`` is zero-initialised
in the reference (transformer.py:224-232), which would hide every temporal
kernel bug, and default-init consumes the global RNG in
construction order.  Instead every tensor is drawn from its own generator seeded
by crc32(key) ^ seed, so any implementation exposing the same state_dict keys
and shapes gets bit-identical weights on any machine with this torch build.
"""
import zlib

import torch


def synth_tensor(key, shape, seed=0):
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    leaf = key.rsplit('.', 1)[-1]
    if 'norm' in key and leaf == 'weight':
        return 1.0 + 0.1 * t
    if leaf == 'bias':
        return 0.02 * t
    if key in ('cls_token', 'pos_embed', 'time_embed', 'mask_token'):
        return 0.02 * t
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return t * (0.7 / max(fan_in, 1) ** 0.5)      # keeps activations O(1) through 12 layers


def synth_state_dict(shapes, seed=0):
    """shapes: {key: shape} (e.g. from a module's state_dict)."""
    return {k: synth_tensor(k, tuple(v), seed) for k, v in shapes.items()}


def shapes_of(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, 'state_dict') else module_or_sd
    return {k: tuple(v.shape) for k, v in sd.items()}


def synth_clip(b, t, c=3, h=224, w=224, seed=0):
    g = torch.Generator().manual_seed(1000 + seed)
    return torch.randn(b, t, c, h, w, generator=g, dtype=torch.float32)
