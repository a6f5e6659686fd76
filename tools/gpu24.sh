python tools/tape_bench.py gpurun_in/p5 poseidon2 65536 8 2>&1 | grep TB
CW_LIB=gpurun_in/libcircom_amd_prof.so python tools/tape_bench.py gpurun_in/p5 poseidon2 65536 2 2>&1 | grep "PROF"
