# round 2 (second session), GPU call 1: kernel tests of the new epilogue modes / continuous flow, per-shape tables, bench A/B
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt or selftest" > $O/r2b1_pytest_kernels.log 2>&1
tail -5 $O/r2b1_pytest_kernels.log
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -k "small or cfg1 or t8_train or autocast" > $O/r2b1_pytest_models.log 2>&1
tail -5 $O/r2b1_pytest_models.log
timeout 300 python tools/gemm_shapes.py 96 8 --old-gelu > $O/r2b1_shapes_cont1.txt 2>&1
VTX_GEMM_PP_CONT=0 timeout 300 python tools/gemm_shapes.py 96 > $O/r2b1_shapes_cont0.txt 2>&1
VTX_GEMM_PP_TOUCH=2 timeout 300 python tools/gemm_shapes.py 96 > $O/r2b1_shapes_touch2.txt 2>&1
VTX_GEMM_PP_TOUCH=6 timeout 300 python tools/gemm_shapes.py 96 > $O/r2b1_shapes_touch6.txt 2>&1
cat $O/r2b1_shapes_cont1.txt
grep -h "sum:\|fwd\|dgrad" $O/r2b1_shapes_cont0.txt | head -30
grep -h "sum:\|(+x)\|g')" $O/r2b1_shapes_touch2.txt $O/r2b1_shapes_touch6.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown > $O/r2b1_bench_default.log 2>&1; tail -1 $O/r2b1_bench_default.log | cut -c1-300
VTX_GEMM_PP_CONT=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown > $O/r2b1_bench_cont0.log 2>&1; tail -1 $O/r2b1_bench_cont0.log | cut -c1-300
