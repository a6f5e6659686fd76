#!/bin/bash
# round 6, twelfth GPU run: the whole GPU suite on the tree with folded boolean rows, smoke, and the lines of configs 4 / 5 with
# the new policies (16 hardware queues + full waves + 32 batches in flight for the Semaphore shard; folded check for the verifier)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1400 python -m pytest tests -m gpu -q --durations=8) > gpurun_out/r06r_gpu_suite.log 2>&1
tail -5 gpurun_out/r06r_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06r_smoke.log 2>&1; tail -2 gpurun_out/r06r_smoke.log
timeout 900 python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 128 --warmup 32 > gpurun_out/r06r_bench_semaphore20p_shard1024.json 2> gpurun_out/r06r_sema.err
timeout 900 python bench.py --workload ecdsa_verify --steps 6 --warmup 3 > gpurun_out/r06r_bench_ecdsa_verify_1024.json 2> gpurun_out/r06r_ecdsa.err
timeout 900 python bench.py --workload poseidon2 > gpurun_out/r06r_bench_poseidon2.json 2> gpurun_out/r06r_poseidon2.err
for f in gpurun_out/r06r_bench_*.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '%.5g' % d['value'], d['ms_per_step'], d['config'].get('in_flight'), d['config'].get('lanes_per_wave'), d['isolated'].get('kernels_ms'), (d.get('parity') or {}).get('parity_checked'), (d.get('cpu_baseline') or {}).get('value'))"; done
