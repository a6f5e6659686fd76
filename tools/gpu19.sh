CW_LIB=gpurun_in/libcircom_amd_nostore.so python tools/tape_bench.py gpurun_in/new/ov4 sha256_512 4096 2 2>&1 | grep "TB\|PROF all\|PROF strand  0"
CW_LIB=gpurun_in/libcircom_amd_nostore.so python tools/tape_bench.py gpurun_in/new/ov4 poseidon2 65536 2 2>&1 | grep "TB\|PROF all\|PROF strand  0"
