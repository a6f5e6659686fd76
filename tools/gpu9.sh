set -x
mkdir -p gpurun_out
rm -f gpurun_out/traffic.json
bash tools/profile.sh r1b_poseidon2 poseidon2:65536 2>&1 | tail -25
bash tools/profile.sh r1b_sha256_512 sha256_512:4096 --workload sha256_512 --batch 4096 2>&1 | tail -25
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json
