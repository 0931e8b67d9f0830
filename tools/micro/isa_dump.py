"""Disassemble a csrc object (container, no GPU) and print per-kernel facts that the resource remarks do not show.

    python tools/micro/isa_dump.py /tmp/gemm_nt.o [name-filter] [--save DIR]

Per kernel: instruction count, scratch instructions and where they sit relative to the first / last MFMA, `s_waitcnt vmcnt(0)`
count, global stores by form (saddr / per-lane 64-bit address).  --save writes one .s file per kernel.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, 'k.o')
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, '--offloading', local], cwd=tmp, capture_output=True, check=False)
        dev = [f for f in os.listdir(tmp) if 'amdgcn' in f]
        r = subprocess.run([OBJDUMP, '-d', os.path.join(tmp, dev[0])], capture_output=True, text=True, check=True)
        return r.stdout.split('\n')
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernels(lines):
    heads = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r'^[0-9a-f]+ <(.*)>:', l)] if m]
    heads.append((len(lines), 'end'))
    for (a, n), (b, _) in zip(heads, heads[1:]):
        yield n, [l.split('//')[0].strip() for l in lines[a + 1:b]]


def main():
    obj = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else ''
    save = sys.argv[sys.argv.index('--save') + 1] if '--save' in sys.argv else None
    if save:
        os.makedirs(save, exist_ok=True)
    for name, body in kernels(disassemble(obj)):
        if filt not in name:
            continue
        short = re.sub(r'^_ZN3vtx\d+', '', name)[:60]
        sc = [i for i, l in enumerate(body) if l.startswith('scratch_')]
        mf = [i for i, l in enumerate(body) if l.startswith('v_mfma')]
        vm0 = sum(1 for l in body if 's_waitcnt' in l and 'vmcnt(0)' in l)
        st_s = sum(1 for l in body if l.startswith('global_store') and re.search(r', s\[\d+:\d+\]', l))
        st_v = sum(1 for l in body if l.startswith('global_store') and ', off' in l)
        print(f'{short:60s} n={len(body):6d} mfma@[{mf[0] if mf else -1},{mf[-1] if mf else -1}] scratch={len(sc)} {sc[:12]} '
              f'vmcnt0={vm0} stores saddr={st_s} vaddr={st_v}')
        if save:
            with open(os.path.join(save, re.sub(r'[^A-Za-z0-9_]', '_', short) + '.s'), 'w') as f:
                f.write('\n'.join(f'{i:6d}  {l}' for i, l in enumerate(body)))


if __name__ == '__main__':
    main()
