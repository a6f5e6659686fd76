set -x
mkdir -p gpurun_out
rm -f gpurun_out/traffic.json
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/profile.sh r1g_poseidon2 poseidon2:65536 2>&1 | tail -25
bash tools/profile.sh r1g_sha256_512 sha256_512:4096 --workload sha256_512 --batch 4096 2>&1 | tail -25
cp gpurun_out/traffic.json profiles/traffic.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json
python bench.py --workload sha256_512 --batch 4096 --steps 5 2>/dev/null | tail -1 > gpurun_out/bench_sha256_512.json; cat gpurun_out/bench_sha256_512.json
python bench.py --workload semaphore20 --batch 65536 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/bench_semaphore20.json; cat gpurun_out/bench_semaphore20.json
python bench.py --workload semaphore20 --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_semaphore20_8192.json; cat gpurun_out/bench_semaphore20_8192.json
python bench.py --workload sha256_512 --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_sha256_512_8192.json; cat gpurun_out/bench_sha256_512_8192.json
