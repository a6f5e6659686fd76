# round 5, GPU call 4: ECDSA verifier with measured scheduler costs + 16-entry D_BITS; default SHA bench (new timing fields, O1 egress of the
# whole batch); register variants of the emitted SHA code under the default step (does a smaller kernel let the ingest of the other batch in?)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
CW_ARTEFACT_FP=r05exp3 timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05d_bench_ecdsa_1024.json 2> gpurun_out/r05d_bench_ecdsa_1024.err
tail -2 gpurun_out/r05d_bench_ecdsa_1024.err
CW_ARTEFACT_FP=r05exp3 timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline --in-flight 1 > gpurun_out/r05d_bench_ecdsa_1024_one_in_flight.json 2> gpurun_out/r05d_bench_ecdsa_1024_one.err
CW_ARTEFACT_FP=r05exp1 timeout 900 python bench.py > gpurun_out/r05d_bench_default.json 2> gpurun_out/r05d_bench_default.err
tail -3 gpurun_out/r05d_bench_default.err
for t in v224 v192; do
  CW_ARTEFACT_FP=$t timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-small --no-parity > gpurun_out/r05d_bench_default_$t.json 2> gpurun_out/r05d_bench_default_$t.err
  tail -2 gpurun_out/r05d_bench_default_$t.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05d_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], "isolated", d["isolated"].get("kernels_ms"), "in_step", d.get("in_step_kernels_ms"), "O1", (d.get("value_canonical_O1") or {}).get("witnesses_per_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
