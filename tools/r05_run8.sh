# round 5, GPU call 8: wave priority by age for 16-strand interpreted schedules (ECDSA verifier), A/B
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
for m in 0 0x80000000 0 0x80000000; do
  if [ "$m" = "0" ]; then unset CW_PRIO_MASK; else export CW_PRIO_MASK=$m; fi
  timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline --no-parity --in-flight 1 > gpurun_out/r05h_ecdsa_prio_$m.json 2> gpurun_out/r05h_ecdsa_prio.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05h_ecdsa_prio_$m.json").read().strip().splitlines()[-1])
print("CW_PRIO_MASK=$m", "value %.5g" % d["value"], d["isolated"]["kernels_ms"])
PY
done
