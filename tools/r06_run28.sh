#!/bin/bash
# round 6, twenty-eighth GPU run: the whole 8 192-instance Semaphore job on one GPU - lanes per wave x batches in flight under 16 queues
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for spec in "32 2" "32 4" "64 2" "64 4" "64 6" "64 8"; do
  set -- $spec
  CW_LANES=$1 timeout 600 python bench.py --workload semaphore20p --batch 8192 --steps 48 --warmup 8 --no-cpu-baseline --no-parity --in-flight $2 > gpurun_out/r06ag_sema8192_l$1_if$2.json 2> gpurun_out/r06ag_sema8192_l$1_if$2.err
  tail -1 gpurun_out/r06ag_sema8192_l$1_if$2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lanes $1 in flight $2:', '%.5g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['config']['engine'][:70], {k: round(v, 3) for k, v in d['isolated']['kernels_ms'].items()})"
done
CW_LANES=64 CW_FP_FUSED=0 timeout 600 python bench.py --workload semaphore20p --batch 8192 --steps 48 --warmup 8 --no-cpu-baseline --no-parity --in-flight 4 > gpurun_out/r06ag_sema8192_l64_if4_unfused.json 2>/dev/null
tail -1 gpurun_out/r06ag_sema8192_l64_if4_unfused.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lanes 64 in flight 4 unfused:', '%.5g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['config']['engine'][:70], {k: round(v, 3) for k, v in d['isolated']['kernels_ms'].items()})"
