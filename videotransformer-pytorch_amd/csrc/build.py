"""Build libvtx.so (gfx950) with hipcc -- no torch headers, no cmake.

    python videotransformer-pytorch_amd/csrc/build.py [--force]

Each .hip file is compiled to an object (skipped when up to date) and linked into
videotransformer-pytorch_amd/libvtx.so.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, 'libvtx.so')
OBJ = os.path.join(HERE, '_obj')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
SOURCES = ['api.hip', 'ln.hip', 'gemm_nt.hip', 'gemm_tn.hip', 'attn.hip', 'attn_mfma.hip', 'elementwise.hip', 'hog.hip', 'optim.hip', 'head.hip', 'mvit.hip', 'wprod.hip', 'xattn_mfma.hip']
# (Round 6 tried -fno-slp-vectorize -- hipcc's SLP pass packs adjacent float32 multiplies / FMAs of the epilogues into v_pk_*
# instructions and pays for it with register shuffles; beside the partner wave's MFMAs a packed VALU instruction costs more than the
# two it replaces (MI355X_MICROARCH.md): step 128.24 -> 127.84 ms over three interleaved A/B rounds.  NOT adopted: without the packing
# the compiler fuses other multiply-add pairs, every bf16 result moves inside its rounding noise, and the noisiest statistic of the
# suite -- the 24-layer default-stream maximum -- landed on the wrong side of its bar.  `build.py --variant noslp -fno-slp-vectorize`.)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result', '-Wno-unused-value', '-Wno-inline-asm',
         '-Wno-cuda-compat']
EXTRA = {'hog.hip': ['-ffp-contract=off']}       # bit-exact HOG: no fma contraction


def _deps():
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')]
    hdrs.append(os.path.join(PKG, '..', 'include', 'vtx.h'))
    hdrs.append(os.path.abspath(__file__))                       # the flags live here
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace('.hip', '.o'))
    spath = os.path.join(HERE, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) > max(os.path.getmtime(spath), _deps())):
        return obj, False
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ['-c', spath, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build_variant(name, defines):
    """An experimental build next to the product library: libvtx_<name>.so with -D<defines> (objects under _obj/<name>/);
    load it with VTX_LIB=<path> to A/B two builds on the same GPU box (tools/micro/*.sh).  Not part of build()."""
    obj_dir = os.path.join(OBJ, name)
    os.makedirs(obj_dir, exist_ok=True)
    out = os.path.join(PKG, f'libvtx_{name}.so')

    def one(src):
        obj = os.path.join(obj_dir, src.replace('.hip', '.o'))
        # (an entry that starts with '-' is passed to hipcc as it is: --variant noslp -fno-slp-vectorize)
        cmd = ([HIPCC] + FLAGS + EXTRA.get(src, []) + [d if d.startswith('-') else '-D' + d for d in defines] +
               ['-c', os.path.join(HERE, src), '-o', obj])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{r.stderr}')
        return obj
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(one, SOURCES))
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stderr}')
    return out


def build(force=False, verbose=True):
    """Compile what is stale and link.  Serialised across processes by a file lock: N ranks of a multi-GPU launch may all
    find the library stale at the same moment (a fresh copy of the tree) and must not write the same objects concurrently."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(OUT):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'built {OUT}')
    elif verbose:
        print(f'{OUT} up to date')
    return OUT


if __name__ == '__main__':
    if '--variant' in sys.argv:                         # python build.py --variant NAME DEFINE[=VALUE] ...
        i = sys.argv.index('--variant')
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        build(force='--force' in sys.argv)
