# round 5, GPU call 2: tier-2 on several strands + native long_div + D_BITS (tests, ECDSA bench), register variants of the emitted SHA code
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ecdsa.py tests/test_functions.py tests/test_witness_list.py tests/test_fpjit.py tests/test_more_circuits.py -m gpu -q --durations=10 > gpurun_out/r05b_tests.log 2>&1
tail -15 gpurun_out/r05b_tests.log
export CW_ARTEFACT_FP=r05exp1
timeout 900 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 > gpurun_out/r05b_bench_ecdsa_1024.json 2> gpurun_out/r05b_bench_ecdsa_1024.err
tail -3 gpurun_out/r05b_bench_ecdsa_1024.err
timeout 600 python bench.py --workload ecdsa_verify --batch 128 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05b_bench_ecdsa_128.json 2> gpurun_out/r05b_bench_ecdsa_128.err
tail -3 gpurun_out/r05b_bench_ecdsa_128.err
for tag in base v128; do
  TAG=$tag ENGINES=jit timeout 300 python tools/jit_bench.py 2048 2097152 5 > gpurun_out/r05b_jit_${tag}_2M.log 2>&1; tail -2 gpurun_out/r05b_jit_${tag}_2M.log
done
TAG=v128 ENGINES=jit timeout 300 python tools/jit_bench.py 2048 4194304 5 > gpurun_out/r05b_jit_v128_4M.log 2>&1; tail -2 gpurun_out/r05b_jit_v128_4M.log
TAG=base ENGINES=jit timeout 300 python tools/jit_bench.py 2048 4194304 5 > gpurun_out/r05b_jit_base_4M.log 2>&1; tail -2 gpurun_out/r05b_jit_base_4M.log
