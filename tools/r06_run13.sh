#!/bin/bash
# round 6, thirteenth GPU run: cw_run_check (a step as one HIP-graph launch): GPU tests, then BASELINE config 3 (Sha256(512) x 4 096,
# 0.12 ms per step with plain launches whatever is in flight) with plain / graph steps, 4 / 16 hardware queues, 4 .. 32 in flight
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_run_check_graph.py -q -m gpu -n 3 > gpurun_out/r06s_graph_tests.log 2>&1
tail -15 gpurun_out/r06s_graph_tests.log
run() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 600 python bench.py $wl --steps 512 --warmup 64 --no-cpu-baseline --no-parity $ARGS > gpurun_out/r06s_$name.json 2> gpurun_out/r06s_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06s_$name.json").read().strip().splitlines()[-1])
    print("$name", "value %.4g" % d["value"], "ms/step %.4f" % d["ms_per_step"], "in_flight", d["config"]["in_flight"], d["config"]["step_launch"][:12], d["isolated"]["kernels_ms"], d["in_step_kernels_ms"])
except Exception as e:
    print("$name unreadable", e)
PY
}
S512="--workload sha256_512 --batch 4096"
ARGS="--graph off" run sha512_plain_q4_if4 "$S512" GPU_MAX_HW_QUEUES=4
ARGS="--graph on" run sha512_graph_q4_if4 "$S512" GPU_MAX_HW_QUEUES=4
ARGS="--graph on --in-flight 8" run sha512_graph_q8_if8 "$S512" GPU_MAX_HW_QUEUES=8
ARGS="--graph on --in-flight 16" run sha512_graph_q16_if16 "$S512" GPU_MAX_HW_QUEUES=16
ARGS="--graph on --in-flight 32" run sha512_graph_q16_if32 "$S512" GPU_MAX_HW_QUEUES=16
ARGS="--graph on --in-flight 32" run sha512_graph_q32_if32 "$S512" GPU_MAX_HW_QUEUES=32
ARGS="--graph off --in-flight 16" run sha512_plain_q16_if16 "$S512" GPU_MAX_HW_QUEUES=16
ARGS="--graph on --in-flight 32" run sema_graph_q16_if32 "--workload semaphore20p --total-batch 8192 --shard-of 8" GPU_MAX_HW_QUEUES=16
ARGS="--graph on" run poseidon2_graph "--workload poseidon2"
ARGS="--graph off" run poseidon2_plain "--workload poseidon2"
