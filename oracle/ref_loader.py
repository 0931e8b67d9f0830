"""Load the UNMODIFIED reference modules from /root/reference (dev container only).

TEST INFRASTRUCTURE ONLY.  ``pytorch_lightning`` and ``pytorchvideo`` are not
installed, so they are stubbed in ``sys.modules`` exactly as SURVEY.md App. B
describes (utils.py:9, video_transformer.py:15-17).  The reference modules are
registered under ``ref_*`` names so they never collide with the drop-in modules
of the same file names in videotransformer-pytorch_amd/.
"""
import importlib
import os
import sys
import types

REF_DIR = os.environ.get('VTX_REFERENCE_DIR', '/root/reference')
_NAMES = ('utils', 'weight_init', 'transformer', 'video_transformer', 'mask_generator')


def available():
    return os.path.isfile(os.path.join(REF_DIR, 'transformer.py'))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _set_attributes(obj, params):
    for k, v in params.items():
        if k != 'self':
            setattr(obj, k, v)


def load():
    """Returns a namespace with .transformer, .video_transformer, .mask_generator."""
    if 'ref_transformer' in sys.modules:
        return types.SimpleNamespace(**{n: sys.modules['ref_' + n] for n in _NAMES})
    if not available():
        raise RuntimeError(f'reference not found under {REF_DIR}')
    if 'pytorch_lightning' not in sys.modules:
        _stub('pytorch_lightning')
        _stub('pytorch_lightning.utilities')
        _stub('pytorch_lightning.utilities.distributed', rank_zero_only=lambda f: f)
    if 'pytorchvideo' not in sys.modules:
        _stub('pytorchvideo')
        _stub('pytorchvideo.layers', MultiScaleBlock=None,
              SpatioTemporalClsPositionalEncoding=None)
        _stub('pytorchvideo.layers.utils', round_width=None, set_attributes=_set_attributes)
        _stub('pytorchvideo.models')
        _stub('pytorchvideo.models.vision_transformers', MultiscaleVisionTransformers=None)
    saved = {n: sys.modules.pop(n) for n in _NAMES if n in sys.modules}
    saved_path = list(sys.path)
    sys.path.insert(0, REF_DIR)
    try:
        mods = {n: importlib.import_module(n) for n in _NAMES}
    finally:
        sys.path[:] = saved_path
        for n in _NAMES:
            m = sys.modules.pop(n, None)
            if m is not None:
                sys.modules['ref_' + n] = m
        sys.modules.update(saved)
    return types.SimpleNamespace(**mods)
