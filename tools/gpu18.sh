for d in gpurun_in/ov0 gpurun_in/ov8 gpurun_in/xc2/ov0 gpurun_in/xc3/ov4 gpurun_in/xc4/ov8 gpurun_in/xc8/ov8; do
  python tools/tape_bench.py $d sha256_512 4096 4 2>&1 | grep TB
  python tools/tape_bench.py $d poseidon2 65536 8 2>&1 | grep TB
done
