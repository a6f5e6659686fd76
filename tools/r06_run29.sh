#!/bin/bash
# round 6, twenty-ninth GPU run: every BASELINE config's line with the final policies of bench.py (default arguments per workload)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --workload semaphore20p --batch 8192 > gpurun_out/r06ah_bench_semaphore20p_8192.json 2> gpurun_out/r06ah_1.err
timeout 900 python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 128 --warmup 32 > gpurun_out/r06ah_bench_semaphore20p_shard1024.json 2> gpurun_out/r06ah_2.err
timeout 900 python bench.py --workload ecdsa_verify --steps 6 --warmup 3 > gpurun_out/r06ah_bench_ecdsa_verify_1024.json 2> gpurun_out/r06ah_3.err
timeout 900 python bench.py --workload ecdsa_verify --total-batch 1024 --shard-of 8 --steps 6 --warmup 3 > gpurun_out/r06ah_bench_ecdsa_verify_shard128.json 2> gpurun_out/r06ah_4.err
timeout 900 python bench.py --workload sha256_512 --batch 4096 --steps 512 --warmup 64 > gpurun_out/r06ah_bench_sha256_512_4096.json 2> gpurun_out/r06ah_5.err
timeout 900 python bench.py --workload poseidon2 --steps 300 --warmup 30 > gpurun_out/r06ah_bench_poseidon2.json 2> gpurun_out/r06ah_6.err
timeout 900 python bench.py --workload poseidon2_goldilocks > gpurun_out/r06ah_bench_poseidon2_goldilocks.json 2> gpurun_out/r06ah_7.err
timeout 900 python bench.py --workload bigmultmodp > gpurun_out/r06ah_bench_bigmultmodp.json 2> gpurun_out/r06ah_8.err
for f in gpurun_out/r06ah_bench_*.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '%.5g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], d['config'].get('in_flight'), d['config'].get('lanes_per_wave'), (d['config'].get('step_launch') or '')[:10], (d.get('isolated') or {}).get('kernels_ms'), (d.get('parity') or {}).get('parity_checked'), (d.get('cpu_baseline') or {}).get('value'))" 2>&1 | tail -1; done
