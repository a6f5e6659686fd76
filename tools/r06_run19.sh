#!/bin/bash
# round 6, nineteenth GPU run: the Semaphore shard with full waves and 16 hardware queues - with 16+ evaluations in flight every CU holds an
# evaluation workgroup and the small kernels of a step (ingest, check) wait for a free CU (check 0.87 ms alone, 7.7 ms in step).
# Fewer batches in flight than queues: some CUs stay free for them?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 8 10 12 13 14 15 16 20 24; do
  CW_LANES=64 timeout 600 python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 256 --warmup 64 --no-cpu-baseline --no-parity --in-flight $n > gpurun_out/r06y_sema_if$n.json 2> gpurun_out/r06y_sema_if$n.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06y_sema_if$n.json").read().strip().splitlines()[-1])
    print("in flight $n", "value %.4g" % d["value"], "ms/step %.4f" % d["ms_per_step"], "lanes", d["config"]["lanes_per_wave"], "in step", {k: round(v, 2) for k, v in d["in_step_kernels_ms"].items()})
except Exception as e:
    print("$n unreadable", e)
PY
done
