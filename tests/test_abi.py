"""The C-ABI library loads and exports every symbol include/vtx.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from helpers import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'vtx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vtx_[a-z0-9_]+)\s*\(', src)))


def _lib_path():
    import __graft_entry__ as ge
    ge.ensure_built()
    return os.path.join(ROOT, 'videotransformer-pytorch_amd', 'libvtx.so')


def test_header_declares_expected_surface():
    names = _declared()
    for must in ['vtx_layernorm_fwd', 'vtx_layernorm_bwd', 'vtx_gemm_nt', 'vtx_gemm_tn', 'vtx_attn_fwd', 'vtx_attn_bwd',
                 'vtx_patch_rows', 'vtx_cls_mean_fwd', 'vtx_hog_fwd', 'vtx_maskfeat_blend_fwd', 'vtx_maskfeat_loss_fwd',
                 'vtx_maskfeat_loss_bwd', 'vtx_version', 'vtx_last_error_string']:
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib_path())
    for name in _declared():
        assert hasattr(lib, name), f'libvtx.so does not export {name}'
    lib.vtx_version.restype = ctypes.c_int
    assert lib.vtx_version() >= 100


def test_python_binding_covers_header():
    from vtx import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    _lib.load()


def test_bad_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any launch: callable on a GPU-less host."""
    from vtx import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    d.dtype, d.M, d.N, d.K = 1, 16, 12, 64                      # N not a multiple of 8
    assert lib.vtx_gemm_nt(ctypes.byref(d), None) == -1
    assert b'multiples of 8' in lib.vtx_last_error_string()
    assert lib.vtx_hog_fwd(None, 1, 224, 224, None, 0, None, None, None) == -1
    a = _lib.AttnDesc()
    a.S, a.L, a.H, a.hd = 1, 8, 2, 48                            # unsupported head dim
    assert lib.vtx_attn_fwd(ctypes.byref(a), None) == -1


def test_struct_layouts_match_ctypes(tmp_path):
    """The descriptor structs cross the boundary by pointer: the ctypes mirrors in vtx/_lib.py must have
    the size and field offsets the C compiler gives include/vtx.h (compiled here as plain C)."""
    import subprocess
    from vtx import _lib
    structs = {'vtx_rowmap': _lib.RowMap, 'vtx_gemm_desc': _lib.GemmDesc, 'vtx_gemm_tn_desc': _lib.GemmTnDesc, 'vtx_wprod_desc': _lib.WprodDesc,
               'vtx_attn_desc': _lib.AttnDesc, 'vtx_attn_bwd_desc': _lib.AttnBwdDesc, 'vtx_mt_tensor': _lib.MtTensor, 'vtx_pool_desc': _lib.PoolDesc, 'vtx_xattn_desc': _lib.XAttnDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "vtx.h")}"',
             'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c99', '-o', str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {}
    for ln in out.splitlines():
        s, f, v = ln.split()
        got[(s, f)] = int(v)
    for cname, cls in structs.items():
        assert got[(cname, 'size')] == ctypes.sizeof(cls), f'{cname}: sizeof differs'
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, f'{cname}.{fname}: offset differs'


def test_integration_stub_matches_the_header(tmp_path):
    """INTEGRATION.md section 2 is the binding a maintainer copies: execute the documented block and hold its RowMap and
    argument list against include/vtx.h as gcc lays it out (round 3 added vtx_rowmap.tab; the stub kept three fields for
    two rounds and would have passed stack garbage as the table pointer)."""
    import subprocess
    from helpers import exec_integration_stub
    _lib_path()
    ns = exec_integration_stub()
    RowMap = ns['RowMap']
    probe = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "vtx.h")}"', 'int main(void) {',
             '  printf("size %zu\\n", sizeof(vtx_rowmap));']
    for fname, _ in RowMap._fields_:
        probe.append(f'  printf("{fname} %zu\\n", offsetof(vtx_rowmap, {fname}));')
    probe += ['  return 0;', '}']
    src = tmp_path / 'rowmap.c'
    src.write_text('\n'.join(probe))
    exe = tmp_path / 'rowmap'
    subprocess.run(['gcc', '-std=c99', '-o', str(exe), str(src)], check=True)
    got = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(got['size']) == ctypes.sizeof(RowMap) == 24
    for fname, _ in RowMap._fields_:
        assert int(got[fname]) == getattr(RowMap, fname).offset, fname
    from vtx import _lib
    assert [f for f, _ in RowMap._fields_] == [f for f, _ in _lib.RowMap._fields_]
    # the documented argument list = the repository's own binding of the same entry point (by ctypes size and kind)
    doc = ns['_vtx'].vtx_layernorm_fwd.argtypes
    own = _lib.SIGNATURES['vtx_layernorm_fwd'][1]
    assert len(doc) == len(own) == 15
    for a, b in zip(doc, own):
        assert ctypes.sizeof(a) == ctypes.sizeof(b), (a, b)
        assert issubclass(a, ctypes.Structure) == issubclass(b, ctypes.Structure), (a, b)


def test_bench_traffic_helper_reads_committed_pmc_passes():
    """bench.py derives roofline.traffic from the committed rocprofv3 PMC dumps of the default command
    (profiles/round2_pmc_{FETCH,WRITE}_SIZE_b<batch>.txt); any other configuration has no counters."""
    import glob
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ['bench.py']
    try:
        spec.loader.exec_module(mod)
        default_batch = mod.parse().batch
    finally:
        sys.argv = argv

    class A:
        precision, frames = 'bf16', 8
    committed = glob.glob(os.path.join(ROOT, 'profiles', f'round2_pmc_FETCH_SIZE_b{default_batch}.txt'))
    t = mod.pmc_traffic_per_launch(default_batch, A)
    if committed:
        assert t is not None and 2e8 < t < 5e9, t        # a few hundred MB .. a few GB per GEMM launch
    else:
        assert t is None
    assert mod.pmc_traffic_per_launch(7, A) is None      # only the profiled configuration has counters


def test_options_api():
    """vtx_set_option: known switches parse, unknown names / values are rejected (no GPU needed)."""
    import vtx
    vtx.set_option('gemm_nt', 'ring256x3')
    vtx.set_option('gemm_nt', 'auto')
    vtx.set_option('pp_grid', '64')
    vtx.set_option('pp_grid', '256')
    for name, value in (('gemm_nt', 'nope'), ('no_such_switch', '1'), ('pp_grid', '7')):
        with pytest.raises(vtx.VtxError):
            vtx.set_option(name, value)


def test_continuous_flow_draw_register_is_private():
    """The persistent GEMM's continuous flow draws tile indices with an inline-asm atomic hipcc does not track; the
    generated code must leave its result register alone until the counted wait (tools/check_isa.py)."""
    import sys
    obj = os.path.join(ROOT, 'videotransformer-pytorch_amd', 'csrc', '_obj', 'gemm_nt.o')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_isa
    if not os.path.exists(obj) or not os.path.exists(check_isa.OBJDUMP):
        pytest.skip('needs the compiled object of csrc/gemm_nt.hip and llvm-objdump')
    assert check_isa.check(obj) == []


def test_one_wave_per_simd_weight_gradient_kernel_has_no_scratch_and_no_fragment_copies():
    """gemm_tn=w4 (csrc/gemm_tn.hip): 505 registers per wave and fragment reads the compiler cannot see complete -- a spill or a
    register copy between a transpose read and its counted wait breaks it silently or drains the LDS-DMA look-ahead (DESIGN 4.2;
    tools/check_isa.py::check_tn_w4)."""
    import sys
    obj = os.path.join(ROOT, 'videotransformer-pytorch_amd', 'csrc', '_obj', 'gemm_tn.o')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_isa
    if not os.path.exists(obj) or not os.path.exists(check_isa.OBJDUMP):
        pytest.skip('needs the compiled object of csrc/gemm_tn.hip and llvm-objdump')
    assert check_isa.check_tn_w4(obj) == []
