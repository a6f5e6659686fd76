# round 5, GPU call 3: ECDSA verifier after the bit-sum folding for wide range checks; per-operator clocks of the 16-strand schedule
set -x
export TMPDIR=/tmp CW_ARTEFACT_FP=r05exp2
mkdir -p gpurun_out
timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05c_bench_ecdsa_1024.json 2> gpurun_out/r05c_bench_ecdsa_1024.err
tail -2 gpurun_out/r05c_bench_ecdsa_1024.err
d=gpurun_in/cache/ecdsa_verify_s16_b1_ma_r05exp2
CW_LIB=gpurun_in/libcircom_amd_prof.so timeout 600 python tools/tape_bench.py $d ecdsa_verify 1024 1 > gpurun_out/r05c_ecdsa_prof.log 2>&1
tail -45 gpurun_out/r05c_ecdsa_prof.log
