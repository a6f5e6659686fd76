#!/bin/bash
# round 6, first GPU run: (1) the loop micro-benchmark (what rolling the emitted code buys), (2) the looped emitted SHA-256 code
# against the straight-line one (prebuilt under gpurun_in/cache with CW_ARTEFACT_FP keys r06a / r06a0), (3) the new GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 gpurun_in/ubench_loop > gpurun_out/r06a_ubench_loop.txt 2>&1
echo "ubench rc=$?"
CW_ARTEFACT_FP=r06a timeout 900 python bench.py --steps 20 > gpurun_out/r06a_bench_loop.json 2> gpurun_out/r06a_bench_loop.err
echo "bench loop rc=$?"
CW_JIT_LOOP=0 CW_ARTEFACT_FP=r06a0 timeout 900 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r06a_bench_line.json 2> gpurun_out/r06a_bench_line.err
echo "bench line rc=$?"
CW_ARTEFACT_FP=r06a timeout 1500 python -m pytest tests/test_bitplane.py tests/test_baseline_configs.py -m gpu -x -q -n 2 \
    -k "beyond_one_chunk or metric_workload or emitted_audit or two_blocks" > gpurun_out/r06a_tests.log 2>&1
echo "tests rc=$?"
tail -5 gpurun_out/r06a_tests.log
python - <<'PY'
import json
for f in ("gpurun_out/r06a_bench_loop.json", "gpurun_out/r06a_bench_line.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.3e ms/step %.2f" % (d["value"], d["ms_per_step"]), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3),
              "iso", d["roofline"].get("isolated", {}).get("frac"), "step", {k: d["step"][k] for k in ("input_frac", "all_traffic_frac", "sum_of_parts_alone_ms")},
              "kernels alone", d["isolated"]["kernels_ms"], "in step", d["in_step_kernels_ms"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
