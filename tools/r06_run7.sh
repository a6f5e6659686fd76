#!/bin/bash
# round 6, seventh GPU run: BASELINE config 5 (secp256k1 ECDSA verification, 2.49 M constraints, 1 024 instances) through EMITTED code
# on 16 strands - the interpreter body call_k (128 VGPRs + private segment, native long_div) and D_BITS steps - against the
# interpreting kernel on the same box; artefacts prebuilt under the key r06m
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
CW_ARTEFACT_FP=r06m timeout 1500 python bench.py --workload ecdsa_verify --steps 6 > gpurun_out/r06m_bench_ecdsa_verify_emitted.json 2> gpurun_out/r06m_bench_ecdsa_verify_emitted.err; echo "emitted rc=$?"; tail -3 gpurun_out/r06m_bench_ecdsa_verify_emitted.err
CW_FP_JIT=0 CW_ARTEFACT_FP=r06m timeout 1500 python bench.py --workload ecdsa_verify --steps 6 --no-cpu-baseline > gpurun_out/r06m_bench_ecdsa_verify_interpreted.json 2> gpurun_out/r06m_bench_ecdsa_verify_interpreted.err; echo "interpreted rc=$?"
CW_ARTEFACT_FP=r06m timeout 1500 python bench.py --workload ecdsa_verify --total-batch 1024 --shard-of 8 --steps 6 --no-cpu-baseline > gpurun_out/r06m_bench_ecdsa_verify_shard128_emitted.json 2> gpurun_out/r06m_bench_ecdsa_verify_shard128_emitted.err; echo "shard rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06m_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06m_bench_")[1], "value %.4g ms/step %.3f in_flight %s" % (d["value"], d["ms_per_step"], d["config"]["in_flight"]), d["config"]["engine"][:60],
              "kernels", d["isolated"]["kernels_ms"], "valu", d["roofline_valu"].get("frac"), (d["roofline_valu"].get("isolated") or {}).get("frac"),
              "parity", (d.get("parity") or {}).get("oracle", "")[:60], (d.get("parity") or {}).get("parity_checked"), "failed", d["failed_instances"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
