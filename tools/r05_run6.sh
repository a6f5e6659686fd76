# round 5, GPU call 6: the audit as emitted code (test + timing at 2^21), D_BITS lower-half stores (ECDSA goldens + bench), default line
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
export CW_ARTEFACT_FP=r05exp5
ln -s ecdsa_verify_s16_b1_ma_r05exp4 gpurun_in/cache/ecdsa_verify_s16_b1_ma_r05exp5
timeout 900 python -m pytest tests/test_bitplane.py tests/test_baseline_configs.py tests/test_ecdsa.py tests/test_golden_wtns.py -m gpu -q --durations=8 > gpurun_out/r05f_tests.log 2>&1
tail -14 gpurun_out/r05f_tests.log
timeout 900 python bench.py > gpurun_out/r05f_bench_default.json 2> gpurun_out/r05f_bench_default.err
tail -3 gpurun_out/r05f_bench_default.err
timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline --in-flight 1 > gpurun_out/r05f_bench_ecdsa_1024_one_in_flight.json 2> gpurun_out/r05f_ecdsa_one.err
timeout 600 python bench.py --workload ecdsa_verify --total-batch 1024 --shard-of 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05f_bench_ecdsa_shard128.json 2> gpurun_out/r05f_ecdsa_shard.err
tail -2 gpurun_out/r05f_ecdsa_shard.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05f_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], "isolated", d["isolated"].get("kernels_ms"), "audit", (d.get("roofline_r1cs") or {}).get("stand_alone_audit"), "one_shot", d.get("one_shot_job"))
    except Exception as e:
        print(f, "ERR", e)
PY
