python tools/tape_bench2.py gpurun_in/s4 poseidon2 65536 20 2>&1 | grep TB2
python tools/tape_bench2.py gpurun_in/s3 poseidon2 65536 20 2>&1 | grep TB2
