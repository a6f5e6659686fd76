#!/bin/bash
# round 6, eighteenth GPU run: every batch in flight on its own part of the chip (CU-masked streams: own hardware queue, own CUs)
# for the Semaphore shard (16 workgroups of 64 instances per batch = 16 CUs), and the fused-check program in that mode
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 256 --warmup 64 --no-cpu-baseline $ARGS > gpurun_out/r06x_$name.json 2> gpurun_out/r06x_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06x_$name.json").read().strip().splitlines()[-1])
    print("$name", "value %.4g" % d["value"], "ms/step %.4f" % d["ms_per_step"], "in_flight", d["config"]["in_flight"], "lanes", d["config"]["lanes_per_wave"], d["config"]["engine"][:50], d["isolated"]["kernels_ms"], d["in_step_kernels_ms"], (d.get("parity") or {}).get("parity_checked"))
except Exception as e:
    print("$name unreadable", e)
    print(open("gpurun_out/r06x_$name.err").read()[-600:])
PY
}
ARGS="" run base CW_X=0
ARGS="--cu-partitions 16" run part16 CW_X=0
ARGS="--cu-partitions 8" run part8 CW_X=0
ARGS="--cu-partitions 32" run part32 CW_X=0
ARGS="" run fused CW_FP_FUSED=1
ARGS="--cu-partitions 16" run fused_part16 CW_FP_FUSED=1
