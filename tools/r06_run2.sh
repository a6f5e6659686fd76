#!/bin/bash
# round 6, second GPU run: (1) the 53-block SHA-256 through the looped emitted code (artefacts prebuilt: tools/r06_prebuild_27008.sh),
# bench line with the reference runtime's golden .wtns inside the batch; (2) the default line with ONE batch in flight;
# (3) the new GPU tests (looped body on Sha256(1024), audit-spill stride, 2^21 ingest, 27008)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
CW_ARTEFACT_FP=r06b timeout 1500 python bench.py --workload sha256_27008 --steps 5 --no-cpu-baseline > gpurun_out/r06b_bench_sha256_27008.json 2> gpurun_out/r06b_bench_sha256_27008.err
echo "bench 27008 rc=$?"; tail -3 gpurun_out/r06b_bench_sha256_27008.err
CW_ARTEFACT_FP=r06a0 timeout 900 python bench.py --steps 20 > gpurun_out/r06b_bench_default.json 2> gpurun_out/r06b_bench_default.err
echo "bench default rc=$?"; tail -3 gpurun_out/r06b_bench_default.err
CW_ARTEFACT_FP=r06a0 timeout 2400 python -m pytest tests/test_bitplane.py tests/test_bitjit.py tests/test_baseline_configs.py -m gpu -x -q -n 3 \
    -k "not config5 and not ecdsa" > gpurun_out/r06b_tests.log 2>&1
echo "tests rc=$?"
tail -5 gpurun_out/r06b_tests.log
python - <<'PY'
import json
for f in ("gpurun_out/r06b_bench_sha256_27008.json", "gpurun_out/r06b_bench_default.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.3e ms/step %.2f" % (d["value"], d["ms_per_step"]), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3),
              "iso", d["roofline"].get("isolated", {}).get("frac"), "step", {k: d["step"][k] for k in ("input_frac", "all_traffic_frac", "sum_of_parts_alone_ms")},
              "kernels alone", d["isolated"]["kernels_ms"], "in step", d["in_step_kernels_ms"], "parity", d["parity"], "in_flight", d["config"]["in_flight"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
