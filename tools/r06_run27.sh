#!/bin/bash
# round 6, twenty-seventh GPU run: config 4's other shapes with this session's policies - the wide Semaphore circuit's shard (semaphore20w), the
# whole 8 192-instance job on one GPU, and what the engine does with 65 536 instances of semaphore20p (single-strand fused program)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --workload semaphore20w --total-batch 8192 --shard-of 8 --steps 96 --warmup 32 > gpurun_out/r06af_bench_semaphore20w_shard1024.json 2> gpurun_out/r06af_w.err
timeout 900 python bench.py --workload semaphore20p --batch 8192 --steps 48 --warmup 8 > gpurun_out/r06af_bench_semaphore20p_8192.json 2> gpurun_out/r06af_p8192.err
timeout 900 python bench.py --workload semaphore20p --batch 65536 --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r06af_bench_semaphore20p_65536.json 2> gpurun_out/r06af_p65536.err
for f in gpurun_out/r06af_bench_*.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '%.5g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['config'].get('in_flight'), d['config'].get('lanes_per_wave'), d['config']['engine'][:60], {k: round(v, 3) for k, v in d['isolated']['kernels_ms'].items()}, (d.get('parity') or {}).get('parity_checked'), (d.get('roofline_valu') or {}).get('frac'))"; done
