#!/bin/bash
# round 6, seventeenth GPU run: BASELINE config 3 (Sha256(512) x 4 096) through the EMITTED bit-plane code (two waves per batch, each
# running the whole circuit: ~1 ms of latency on two SIMDs) with many batches in flight, against the interpreter (256 waves, 0.3 ms)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload sha256_512 --batch 4096 --steps 1024 --warmup 128 --no-cpu-baseline $ARGS > gpurun_out/r06w_$name.json 2> gpurun_out/r06w_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06w_$name.json").read().strip().splitlines()[-1])
    print("$name", "value %.4g" % d["value"], "ms/step %.4f" % d["ms_per_step"], "in_flight", d["config"]["in_flight"], d["config"]["step_launch"][:12], d["config"]["engine"][:40], d["isolated"]["kernels_ms"], (d.get("parity") or {}).get("parity_checked"))
except Exception as e:
    print("$name unreadable", e)
PY
}
ARGS="--in-flight 16" run interp_if16 CW_X=0
ARGS="--in-flight 16" run jit_if16 CW_BITS_JIT=1
ARGS="--in-flight 32" run jit_if32 CW_BITS_JIT=1
ARGS="--in-flight 64" run jit_if64 CW_BITS_JIT=1
ARGS="--in-flight 64" run jit_q32_if64 CW_BITS_JIT=1 GPU_MAX_HW_QUEUES=32
