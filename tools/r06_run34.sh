#!/bin/bash
# round 6, thirty-fourth GPU run: a step of the 256-bit engine as three launches (prologue = init + ingest + reset of the finding words; the
# evaluation; the check, whose first-chunk workgroups merge the fused findings) instead of six - GPU suite, then same-box A/B against the
# previous library (CW_LIB) on the lines that keep many batches in flight
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1400 python -m pytest tests -x -q -m gpu) > gpurun_out/r06am_gpu_suite.log 2>&1
grep -E "passed|failed" gpurun_out/r06am_gpu_suite.log | tail -1
PREV=$PWD/gpurun_in/lib_prev/libcircom_amd.so
run() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 900 python bench.py $wl --no-cpu-baseline --no-parity > gpurun_out/r06am_$name.json 2> gpurun_out/r06am_$name.err
  tail -1 gpurun_out/r06am_$name.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', 'value %.5g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], d['config']['in_flight'], 'in step', {k: round(v, 3) for k, v in d['in_step_kernels_ms'].items()})"
}
S="--workload semaphore20p --total-batch 8192 --shard-of 8 --steps 256 --warmup 64"
for k in 1 2; do
  run sema_new_$k "$S" CW_X=0
  run sema_prev_$k "$S" CW_LIB=$PREV
done
S8="--workload semaphore20p --batch 8192 --steps 48 --warmup 8"
run sema8192_new "$S8" CW_X=0
run sema8192_prev "$S8" CW_LIB=$PREV
P="--workload poseidon2 --steps 300 --warmup 30"
run poseidon_new "$P" CW_X=0
run poseidon_prev "$P" CW_LIB=$PREV
B="--workload bigmultmodp --steps 200 --warmup 20"
run bigmult_new "$B" CW_X=0
run bigmult_prev "$B" CW_LIB=$PREV
