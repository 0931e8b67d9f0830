# GPU call 3: residual-block flow, direct parameter gradients (hooks counted once)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt or selftest" > $O/r2b3_pytest_kernels.log 2>&1
tail -15 $O/r2b3_pytest_kernels.log
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_dp.py -x -q -k "small or direct or recompute or rccl_one or bench_runs or t8_train" > $O/r2b3_pytest_models.log 2>&1
tail -8 $O/r2b3_pytest_models.log
timeout 300 python tools/gemm_shapes.py 96 8 > $O/r2b3_shapes.txt 2>&1
cat $O/r2b3_shapes.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown > $O/r2b3_bench_default.log 2>&1; tail -1 $O/r2b3_bench_default.log | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-direct-grads > $O/r2b3_bench_nodirect.log 2>&1; tail -1 $O/r2b3_bench_nodirect.log | cut -c1-300
