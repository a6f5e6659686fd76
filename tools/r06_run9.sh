#!/bin/bash
# round 6, ninth GPU run: boolean rows folded into the sums that read the same bits (cw_r1cs_plan.h build_stream):
# GPU tests of the check, then the ECDSA verifier's / the Semaphore shard's check with and without folding on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_r1cs_plan.py -q -m gpu -k "r1cs" -n 4 > gpurun_out/r06o_r1cs_tests.log 2>&1
tail -3 gpurun_out/r06o_r1cs_tests.log
for v in fold nofold; do
  for wl in ecdsa_verify semaphore20p; do
    extra=""; [ $wl = semaphore20p ] && extra="--total-batch 8192 --shard-of 8"
    if [ $v = nofold ]; then export CW_R1CS_NO_FOLD=1; else unset CW_R1CS_NO_FOLD; fi
    timeout 900 python bench.py --workload $wl $extra --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline > gpurun_out/r06o_${wl}_$v.json 2> gpurun_out/r06o_${wl}_$v.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06o_${wl}_$v.json").read().strip().splitlines()[-1])
    print("$wl $v", d["isolated"]["kernels_ms"], "ms/step %.2f" % d["ms_per_step"], "parity", d.get("parity"))
except Exception as e:
    print("$wl $v unreadable", e)
PY
  done
done
