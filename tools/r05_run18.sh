# round 5, GPU call 18 (the last minutes): bench lines of the other BASELINE configs on the final tree
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 100 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05r_bench_ecdsa_verify_1024.json 2>/dev/null
timeout 45 python bench.py --workload poseidon2 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r05r_bench_poseidon2.json 2>/dev/null
timeout 50 python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r05r_bench_semaphore20p_shard1024.json 2>/dev/null
timeout 40 python bench.py --workload sha256_512 --batch 4096 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r05r_bench_sha256_512_4096.json 2>/dev/null
for f in gpurun_out/r05r_bench_ecdsa* gpurun_out/r05r_bench_pos* gpurun_out/r05r_bench_sem* gpurun_out/r05r_bench_sha256_512*; do tail -1 $f | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$f', '%.5g' % d['value'], d['ms_per_step'], d['isolated'].get('kernels_ms'), (d.get('parity') or {}).get('parity_checked'))
except Exception as e: print('$f', 'no line', e)"; done
