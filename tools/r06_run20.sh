#!/bin/bash
# round 6, twentieth GPU run: Poseidon(2) x 65 536 (BASELINE config 1) - batches in flight under 16 hardware queues (round 4 chose three under the
# runtime's 4 queues: 2 -> 59.2 M, 3 -> 62.1 M, 4 -> 58.4 M)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 2 3 4 5 6 8; do
  timeout 600 python bench.py --workload poseidon2 --steps 300 --warmup 30 --no-cpu-baseline --no-parity --in-flight $n > gpurun_out/r06z_poseidon2_if$n.json 2> gpurun_out/r06z_poseidon2_if$n.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06z_poseidon2_if$n.json").read().strip().splitlines()[-1])
    print("in flight $n", "value %.4g" % d["value"], "ms/step %.4f" % d["ms_per_step"], d["config"]["step_launch"][:12], "queues", d["config"]["hw_queues"], "in step", {k: round(v, 3) for k, v in d["in_step_kernels_ms"].items()}, "valu frac", d.get("roofline_valu", {}).get("frac"))
except Exception as e:
    print("$n unreadable", e)
PY
done
GPU_MAX_HW_QUEUES=4 timeout 600 python bench.py --workload poseidon2 --steps 300 --warmup 30 --no-cpu-baseline --no-parity --in-flight 3 > gpurun_out/r06z_poseidon2_q4_if3.json 2>/dev/null
tail -1 gpurun_out/r06z_poseidon2_q4_if3.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('4 queues, 3 in flight', '%.4g' % d['value'], d['ms_per_step'])"
