# round 5, first GPU call: (1) the new GPU tests (native long_div, witness-list egress), (2) per-operator clocks of the ECDSA verifier's
# single-strand schedule on the profiling build (tools/profile_ops.sh), 64 instances
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ecdsa.py tests/test_witness_list.py -m gpu -x -q --durations=10 > gpurun_out/r05a_tests.log 2>&1
tail -5 gpurun_out/r05a_tests.log
gunzip -k gpurun_in/ecdsa_s1/ecdsa_verify.r1cs.gz
CW_LIB=gpurun_in/libcircom_amd_prof.so timeout 900 python tools/tape_bench.py gpurun_in/ecdsa_s1 ecdsa_verify 64 1 > gpurun_out/r05a_ecdsa_prof.log 2>&1
tail -30 gpurun_out/r05a_ecdsa_prof.log
