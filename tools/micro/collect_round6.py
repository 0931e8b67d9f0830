"""Copy the evidence set of tools/micro/profile_round6.sh from gpurun_out/ into profiles/round6_* and print the numbers DESIGN.md 6 quotes.

    python tools/micro/collect_round6.py <git head of the run>
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def families(path):
    rows = list(csv.DictReader(open(path)))
    steps = [int(r['calls']) for r in rows if 'mt_step_kernel' in r['kernel']][0]
    fam = {}
    for r in rows:
        k, t = r['kernel'], int(r['total_ns']) / 1e6 / steps
        n = ('NT' if 'gemm_nt' in k else 'TN' if 'gemm_tn' in k else 'slab sums' if 'reduce_partials' in k else 'attention' if 'attn_' in k
             else 'LayerNorm' if 'ln_' in k else 'wprod' if 'wprod' in k else 'hog' if 'hog' in k else 'other vtx' if 'vtx::' in k else 'ATen / runtime')
        fam[n] = fam.get(n, 0) + t
    return sum(fam.values()), fam


def main():
    head = sys.argv[1] if len(sys.argv) > 1 else '?'
    for name in ('bench_default_b96', 'bench_force_dp_b96', 'bench_exact_stream_b96', 'bench_exact_grad_stream_b96'):
        with open(os.path.join(P, f'round6_{name}.json'), 'w') as f:
            f.write(open(os.path.join(O, f'r6_{name}.log')).read().strip().splitlines()[-1] + '\n')
    for name in ('rocprofv3_kernel_stats_b96.csv', 'rocprofv3_kernel_stats_exact_b96.csv', 'rocprofv3_kernel_stats_exact_grad_b96.csv', 'pmc_FETCH_SIZE_b96.txt',
                 'pmc_WRITE_SIZE_b96.txt', 'pmc_MFMA_BUSY_b96.txt', 'gpu_suite.log', 'parity_report.txt', 'hog.txt', 'other_configs.txt', 'maskfeat.txt'):
        shutil.copy(os.path.join(O, 'r6_' + name), os.path.join(P, 'round6_' + name))
    with open(os.path.join(P, 'round6_tree.txt'), 'w') as f:
        f.write(f'HEAD {head} (git), evidence run of tools/micro/profile_round6.sh on one box:\n' + open(os.path.join(O, 'r6_tree.txt')).read())
    d = last_json(os.path.join(O, 'r6_bench_default_b96.log'))
    print(f"default: {d['value']:.1f} clips/s, {d['ms_per_step']:.1f} ms, roofline.frac {d['roofline']['frac']:.4f} ({d['roofline']['avg_launch_us']:.1f} us, "
          f"{d['roofline']['achieved']:.0f} TF/s), TN {d['gemm_tn_roofline']['frac']:.3f}, mfma whole step {d['mfma_frac_whole_step']:.3f} / nominal {d['mfma_frac_whole_step_nominal']:.3f}, "
          f"traffic {d['roofline']['traffic']}")
    for o in d.get('other_configs', []):
        print(f"   {o['workload'][:110]}: {o.get('clips_per_s')} clips/s, {o.get('ms_per_step')} ms")
    for name in ('bench_force_dp_b96', 'bench_exact_stream_b96', 'bench_exact_grad_stream_b96'):
        e = last_json(os.path.join(O, f'r6_{name}.log'))
        print(f"{name}: {e['value']:.1f} clips/s, {e['ms_per_step']:.1f} ms, frac {e['roofline']['frac']:.4f}")
    for name in ('rocprofv3_kernel_stats_b96.csv', 'rocprofv3_kernel_stats_exact_b96.csv', 'rocprofv3_kernel_stats_exact_grad_b96.csv'):
        tot, fam = families(os.path.join(O, 'r6_' + name))
        print(f'{name}: {tot:.1f} ms per step: ' + ', '.join(f'{k} {v:.2f} ({100 * v / tot:.1f} %)' for k, v in sorted(fam.items(), key=lambda kv: -kv[1])))
    print(open(os.path.join(O, 'r6_gpu_suite.log')).read().strip().splitlines()[-7][:200])
    rep = open(os.path.join(O, 'r6_parity_report.txt')).read().splitlines()
    print(f'parity lines {len(rep)}, FAIL {sum(l.startswith("FAIL") for l in rep)}')
    print(open(os.path.join(O, 'r6_hog.txt')).read().strip())
    print(open(os.path.join(O, 'r6_maskfeat.txt')).read().strip().splitlines()[-1][:200])


if __name__ == '__main__':
    main()
