#!/bin/bash
# round 6, eleventh GPU run: run 10 showed 8 .. 32 batches in flight all give 3.9 ms per 1 024-instance Semaphore batch whatever the
# lanes per wave: the HIP runtime maps streams onto 4 hardware queues (GPU_MAX_HW_QUEUES), so 4 launches run side by side.
# More queues + full waves (64 instances per workgroup, 16 CUs per batch)?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 600 python bench.py $wl --steps 64 --warmup 16 --no-cpu-baseline --no-parity $ARGS > gpurun_out/r06q_$name.json 2> gpurun_out/r06q_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06q_$name.json").read().strip().splitlines()[-1])
    print("$name", "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "in_flight", d["config"]["in_flight"], d["isolated"]["kernels_ms"])
except Exception as e:
    print("$name unreadable", e)
PY
}
SEMA="--workload semaphore20p --total-batch 8192 --shard-of 8"
ARGS="--in-flight 8" run sema_q8_l16_if8 "$SEMA" GPU_MAX_HW_QUEUES=8
ARGS="--in-flight 16" run sema_q8_l32_if16 "$SEMA" GPU_MAX_HW_QUEUES=8 CW_LANES=32
ARGS="--in-flight 16" run sema_q16_l64_if16 "$SEMA" GPU_MAX_HW_QUEUES=16 CW_LANES=64
ARGS="--in-flight 32" run sema_q16_l64_if32 "$SEMA" GPU_MAX_HW_QUEUES=16 CW_LANES=64
ARGS="--in-flight 32" run sema_q32_l64_if32 "$SEMA" GPU_MAX_HW_QUEUES=32 CW_LANES=64
ARGS="--in-flight 16" run sema_q16_l16_if16 "$SEMA" GPU_MAX_HW_QUEUES=16
S512="--workload sha256_512 --batch 4096"
ARGS="" run sha512_default "$S512" CW_X=0
ARGS="--in-flight 8" run sha512_q8_if8 "$S512" GPU_MAX_HW_QUEUES=8
ARGS="--in-flight 16" run sha512_q16_if16 "$S512" GPU_MAX_HW_QUEUES=16
ARGS="--in-flight 3" run ecdsa_q8_if3 "--workload ecdsa_verify" GPU_MAX_HW_QUEUES=8
