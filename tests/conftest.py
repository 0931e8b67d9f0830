import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'videotransformer-pytorch_amd')
for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


_OPTION_DEFAULTS = dict(gemm_nt='auto', gemm_tn='auto', gemm_nodma='0', tn_safe='0', tn_cus='256', attn_valu='0', attn_hw_fwd='16', attn_hw_bwd='4', pp_grid='256',
                        pp_cg='0', pp_epi='0', pp_cont='1', attn_fused='2', attn_fwd_stream='1', attn_dkv='3', ln_rows='3')


@pytest.fixture
def vtx_opts():
    """set(name, value) -> vtx.set_option; every touched switch is put back to its default afterwards."""
    import vtx
    touched = set()

    def set_(name, value):
        vtx.set_option(name, value)
        touched.add(name)
    yield set_
    for name in touched:
        vtx.set_option(name, _OPTION_DEFAULTS[name])
