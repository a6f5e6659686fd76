# Round-4 bench lines of every BASELINE config on one MI355X (gpurun box), into gpurun_out/r04d_*.json
set -x
mkdir -p gpurun_out
python bench.py > gpurun_out/r04d_bench_sha256_2048_2M.json 2> gpurun_out/r04d_default.err
python bench.py --workload poseidon2 --steps 20 --warmup 3 > gpurun_out/r04d_bench_poseidon2.json 2>/dev/null
python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 10 --warmup 2 > gpurun_out/r04d_bench_semaphore20p_shard1024.json 2>/dev/null
python bench.py --workload semaphore20p --steps 10 --warmup 2 > gpurun_out/r04d_bench_semaphore20p_8192.json 2>/dev/null
python bench.py --workload semaphore20w --total-batch 8192 --shard-of 8 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04d_bench_semaphore20w_shard1024.json 2>/dev/null
python bench.py --workload bigmultmodp --steps 10 --warmup 2 > gpurun_out/r04d_bench_bigmultmodp_8192.json 2>/dev/null
python bench.py --workload sha256_512 --batch 4096 --steps 20 --warmup 3 > gpurun_out/r04d_bench_sha256_512_4096.json 2>/dev/null
# (ecdsa_verify: measured separately, profiles/r04d_bench_ecdsa_verify_1024.json - 2.5 min of box time)
for f in gpurun_out/r04d_bench_*.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', d['value'], d['isolated'], (d.get('parity') or {}).get('parity_checked'), (d.get('cpu_baseline') or {}).get('value'))"; done
