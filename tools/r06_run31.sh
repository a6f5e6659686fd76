#!/bin/bash
# round 6, thirty-first GPU run: lines whose defaults changed last (Goldilocks: six in flight; the verifier's 128-instance shard: as many in
# flight as the HBM holds, up to 16)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --workload ecdsa_verify --total-batch 1024 --shard-of 8 --steps 12 --warmup 4 > gpurun_out/r06aj_bench_ecdsa_verify_shard128.json 2> gpurun_out/r06aj_1.err
timeout 900 python bench.py --workload poseidon2_goldilocks > gpurun_out/r06aj_bench_poseidon2_goldilocks.json 2> gpurun_out/r06aj_2.err
for f in gpurun_out/r06aj_bench_*.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '%.5g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], d['config'].get('in_flight'), d['config'].get('lanes_per_wave'), (d.get('isolated') or {}).get('kernels_ms'), (d.get('parity') or {}).get('parity_checked'))"; done
