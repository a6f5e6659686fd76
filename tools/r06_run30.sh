#!/bin/bash
# round 6, thirtieth GPU run: Poseidon(2) on Goldilocks x 65 536 - batches in flight under 16 queues
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 1 2 3 4 6; do
  timeout 600 python bench.py --workload poseidon2_goldilocks --steps 200 --warmup 20 --no-cpu-baseline --no-parity --in-flight $n > gpurun_out/r06ai_gold_if$n.json 2> gpurun_out/r06ai_gold_if$n.err
  tail -1 gpurun_out/r06ai_gold_if$n.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('in flight $n:', '%.5g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], (d.get('isolated') or {}).get('kernels_ms'))"
done
