"""ISA checks of the built persistent NT GEMM (continuous-flow kernels), run by tests/test_abi.py.

The continuous flow draws its tile indices with an inline-asm atomic whose result register hipcc does not track
(csrc/gemm_nt.hip, "The draw"): correctness rests on two properties of the generated code that this script verifies
in the disassembly of csrc/_obj/gemm_nt.o:
  1. the atomic's destination register has exactly one reader before it is overwritten -- the ds_write that hands the
     index over -- and that reader sits behind the `s_waitcnt vmcnt(6)` with at least six LDS-DMA requests between
     the atomic and the wait (vector memory operations return in order, so the atomic has then returned);
  2. the kernel has no scratch (spill) traffic and no `s_waitcnt vmcnt(0)` between the atomic and the hand-over.

    python tools/check_isa.py [path/to/gemm_nt.o]      -> prints one line per kernel, exit code 1 on a violation
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, 'k.o')
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, '--offloading', local], cwd=tmp, capture_output=True, check=False)
        dev = [f for f in os.listdir(tmp) if 'amdgcn' in f]
        if not dev:
            raise RuntimeError('no device code object found in ' + obj)
        r = subprocess.run([OBJDUMP, '-d', os.path.join(tmp, dev[0])], capture_output=True, text=True, check=True)
        return r.stdout.split('\n')
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernels(lines):
    heads = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r'^[0-9a-f]+ <(.*)>:', l)] if m]
    heads.append((len(lines), 'end'))
    for (a, n), (b, _) in zip(heads, heads[1:]):
        yield n, [l.split('//')[0] for l in lines[a:b]]


def touches(text, num):
    if re.search(r'\bv%d\b' % num, text):
        return True
    return any(int(m.group(1)) <= num <= int(m.group(2)) for m in re.finditer(r'v\[(\d+):(\d+)\]', text))


def check(obj):
    problems, seen = [], 0
    for name, body in kernels(disassemble(obj)):
        if 'gemm_nt_bf16_pp_kernel' not in name:
            continue
        draws = [i for i, l in enumerate(body) if 'global_atomic_add' in l and ', off sc0' in l]
        if not draws:
            continue
        seen += 1
        short = name[name.index('pp_kernel'):][:40]
        if any('scratch_' in l for l in body):
            problems.append(f'{short}: scratch traffic')
        for i0 in draws:
            num = int(re.search(r'global_atomic_add v(\d+),', body[i0]).group(1))
            dma, waited, user = 0, None, None
            for i in range(i0 + 1, len(body)):
                t = body[i]
                if 'global_load_lds' in t and waited is None:
                    dma += 1
                if 's_waitcnt vmcnt(0)' in t and user is None:
                    problems.append(f'{short}: vmcnt(0) between the draw and its hand-over (line {i})')
                if 's_waitcnt vmcnt(6)' in t and waited is None:
                    waited = i
                if touches(t, num):
                    user = (i, t.strip())
                    break
            if user is None or not user[1].startswith('ds_write_b32') or waited is None or waited > user[0]:
                problems.append(f'{short}: draw register v{num} first touched by {user} (wait at {waited})')
            elif dma < 6:
                problems.append(f'{short}: only {dma} DMA requests between the draw and its wait')
            else:
                print(f'{short}: draw v{num} -> {user[1]} behind vmcnt(6), {dma} DMA requests in between: ok')
    if seen == 0:
        problems.append('no continuous-flow kernel found')
    return problems


def check_tn_w4(obj):
    """The one-wave-per-SIMD weight-gradient kernel (csrc/gemm_tn.hip, gemm_tn=w4) depends on three properties hipcc does not
    guarantee: no scratch traffic (a spilled value is reloaded behind `s_waitcnt vmcnt(0)`, which drains the whole
    LDS-DMA look-ahead), the transpose reads' 64-bit halves allocated as the aligned 128-bit tuples the MFMAs take (no v_mov
    between a read and its use: a copy before the counted wait would move stale data), and the phase structure itself: 192
    MFMAs (three copies of the K tile: the pair in the loop + the odd one), 32 + 3 x 64 transpose reads."""
    problems, seen = [], 0
    for name, body in kernels(disassemble(obj)):
        if 'gemm_tn_bf16_w4_kernel' not in name:
            continue
        seen += 1
        text = [l for l in body if l.strip()]
        n_scratch = sum(1 for l in text if 'scratch_' in l)
        n_mfma = sum(1 for l in text if 'v_mfma_f32_32x32x16_bf16' in l)
        n_tr = sum(1 for l in text if 'ds_read_b64_tr_b16' in l)
        idx = [i for i, l in enumerate(text) if 'v_mfma_f32_32x32x16_bf16' in l]
        # inside the K-tile copies (consecutive MFMAs at most 200 lines apart; between the loop and the odd tile the allocator
        # moves the 256 accumulators once per launch, which is harmless)
        inside = [l for a, b in zip(idx, idx[1:]) if b - a <= 200 for l in text[a:b]]
        n_mov64 = sum(1 for l in inside if 'v_mov_b64' in l)
        n_acc_mov = sum(1 for l in inside if 'v_accvgpr' in l)
        ok = n_scratch == 0 and n_mfma == 192 and n_tr == 224 and n_mov64 == 0 and n_acc_mov == 0
        print(f'{name[:48]}: scratch {n_scratch}, mfma {n_mfma}, transpose reads {n_tr}, v_mov_b64 / accvgpr moves inside the K-tile copies '
              f'{n_mov64} / {n_acc_mov}: {"ok" if ok else "VIOLATION"}')
        if not ok:
            problems.append(name)
    if seen != 2:
        problems.append(f'expected two gemm_tn_bf16_w4_kernel instantiations, found {seen}')
    return problems


if __name__ == '__main__':
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'videotransformer-pytorch_amd', 'csrc', '_obj', 'gemm_nt.o')
    bad = check(obj)
    if len(sys.argv) <= 1:
        bad += check_tn_w4(os.path.join(ROOT, 'videotransformer-pytorch_amd', 'csrc', '_obj', 'gemm_tn.o'))
    for b in bad:
        print('VIOLATION', b)
    sys.exit(1 if bad else 0)
