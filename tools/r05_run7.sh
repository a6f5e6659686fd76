# round 5, GPU call 7: ingest kernel with whole-line loads (A/B against round 3's pattern), correctness on the bit-plane tests
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bitplane.py tests/test_sha256.py tests/test_bitjit.py -m gpu -q > gpurun_out/r05g_tests.log 2>&1
tail -4 gpurun_out/r05g_tests.log
for v in 1 0 1 0; do
  CW_INGEST=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-small --no-parity > gpurun_out/r05g_bench_ingest$v.json 2> gpurun_out/r05g_bench_ingest$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05g_bench_ingest$v.json").read().strip().splitlines()[-1])
print("CW_INGEST=$v", "value %.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], d["isolated"]["kernels_ms"], d["in_step_kernels_ms"], d["roofline"]["frac"])
PY
done
