#!/bin/bash
# round 6, tenth GPU run: the Semaphore shard (1 024 instances per batch) with full waves (64 instances per workgroup = 16 CUs per
# batch) and enough batches in flight to fill the chip, against the default (16 lanes per wave, 64 workgroups per batch, 8 in flight)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env..., -- args
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 64 --warmup 16 --no-cpu-baseline --no-parity $ARGS > gpurun_out/r06p_sema_$name.json 2> gpurun_out/r06p_sema_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06p_sema_$name.json").read().strip().splitlines()[-1])
    print("$name", "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "in_flight", d["config"]["in_flight"], d["roofline"].get("lanes_per_workgroup"), d["isolated"]["kernels_ms"])
except Exception as e:
    print("$name unreadable", e)
PY
}
ARGS="" run default CW_X=0
ARGS="--in-flight 16" run l16_if16 CW_X=0
ARGS="--in-flight 8" run l64_if8 CW_LANES=64
ARGS="--in-flight 16" run l64_if16 CW_LANES=64
ARGS="--in-flight 24" run l64_if24 CW_LANES=64
ARGS="--in-flight 32" run l64_if32 CW_LANES=64
ARGS="--in-flight 16" run l32_if16 CW_LANES=32
ARGS="--in-flight 3" ; env timeout 900 python bench.py --workload ecdsa_verify --steps 6 --warmup 3 --no-cpu-baseline --in-flight 3 > gpurun_out/r06p_ecdsa_if3.json 2> gpurun_out/r06p_ecdsa_if3.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06p_ecdsa_if3.json").read().strip().splitlines()[-1])
print("ecdsa if3", "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], d["isolated"]["kernels_ms"])
PY
