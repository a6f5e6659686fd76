#!/bin/bash
# round 6, twenty-second GPU run: the light build of the stream check kernel (eight waves per SIMD, two wires ahead) - GPU tests of
# the check on both builds, then the ECDSA verifier's check alone and with three batches in flight, light against plain on one box;
# the Semaphore shard (a system of products) on both builds as the control
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_run_check_graph.py -q -m gpu -k "r1cs or graph" -n 4 > gpurun_out/r06aa_r1cs_tests.log 2>&1
tail -4 gpurun_out/r06aa_r1cs_tests.log
run() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 900 python bench.py $wl --no-cpu-baseline $ARGS > gpurun_out/r06aa_$name.json 2> gpurun_out/r06aa_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06aa_$name.json").read().strip().splitlines()[-1])
    print("$name", "value %.5g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "in_flight", d["config"]["in_flight"], "alone", {k: round(v, 3) for k, v in d["isolated"]["kernels_ms"].items()}, "in step", {k: round(v, 3) for k, v in d["in_step_kernels_ms"].items()}, "parity", (d.get("parity") or {}).get("parity_checked"))
except Exception as e:
    print("$name unreadable", e)
PY
}
E="--workload ecdsa_verify --steps 6 --warmup 3"
ARGS="--in-flight 3" run ecdsa_light_if3 "$E" CW_R1CS_LIGHT=1
ARGS="--in-flight 3" run ecdsa_plain_if3 "$E" CW_R1CS_LIGHT=0
ARGS="--in-flight 1 --no-parity" run ecdsa_light_if1 "$E" CW_R1CS_LIGHT=1
ARGS="--in-flight 1 --no-parity" run ecdsa_plain_if1 "$E" CW_R1CS_LIGHT=0
S="--workload semaphore20p --total-batch 8192 --shard-of 8 --steps 128 --warmup 32 --no-parity"
ARGS="" run sema_light "$S" CW_R1CS_LIGHT=1
ARGS="" run sema_plain "$S" CW_R1CS_LIGHT=0
