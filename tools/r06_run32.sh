#!/bin/bash
# round 6, thirty-second GPU run: the stream check with TWO wires in flight beyond the current one (no register cap this time: 3 waves per SIMD
# instead of 4) - tests, then the ECDSA verifier alone / three in flight and the Semaphore shard, one against two on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "r1cs" -n 4 2>&1 | tail -1
run() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 900 python bench.py $wl --no-cpu-baseline --no-parity $ARGS > gpurun_out/r06ak_$name.json 2> gpurun_out/r06ak_$name.err
  tail -1 gpurun_out/r06ak_$name.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', 'value %.5g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'alone', {k: round(v, 3) for k, v in d['isolated']['kernels_ms'].items()}, 'in step', {k: round(v, 3) for k, v in d['in_step_kernels_ms'].items()})"
}
E="--workload ecdsa_verify --steps 6 --warmup 3"
ARGS="--in-flight 3" run ecdsa_ahead2_if3 "$E" CW_R1CS_AHEAD=2
ARGS="--in-flight 3" run ecdsa_ahead1_if3 "$E" CW_R1CS_AHEAD=1
ARGS="--in-flight 3" run ecdsa_ahead2_if3_b "$E" CW_R1CS_AHEAD=2
ARGS="--in-flight 3" run ecdsa_ahead1_if3_b "$E" CW_R1CS_AHEAD=1
S="--workload semaphore20p --total-batch 8192 --shard-of 8 --steps 128 --warmup 32"
ARGS="" run sema_ahead2 "$S" CW_R1CS_AHEAD=2
ARGS="" run sema_ahead1 "$S" CW_R1CS_AHEAD=1
