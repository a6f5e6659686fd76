#!/bin/bash
# round 6, closing run: the whole GPU suite, the driver's bench line, its rocprofv3 kernel trace + PMC passes (tools/profile.sh),
# and one line per BASELINE configuration / extra workload on the final source
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06h_gpu_suite.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/r06h_gpu_suite.log
timeout 900 python bench.py > gpurun_out/r06h_bench_sha256_2048_2M.json 2> gpurun_out/r06h_bench_sha256_2048_2M.err
echo "bench default rc=$?"
timeout 1500 bash tools/profile.sh r06h sha256_2048:2097152 > gpurun_out/r06h_profile.log 2>&1
echo "profile rc=$?"; cd $R
for f in $(find gpurun_out/prof_r06h/trace -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r06h_sha256_2048_2M_kernel_stats.csv; done
cp gpurun_out/prof_r06h/summary.txt gpurun_out/r06h_sha256_2048_2M_summary.txt 2>/dev/null
timeout 600 python bench.py --workload poseidon2 > gpurun_out/r06h_bench_poseidon2.json 2> gpurun_out/r06h_bench_poseidon2.err; echo "poseidon2 rc=$?"
timeout 600 python bench.py --workload sha256_512 --batch 4096 > gpurun_out/r06h_bench_sha256_512_4096.json 2> gpurun_out/r06h_bench_sha256_512_4096.err; echo "sha256_512 rc=$?"
timeout 900 python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 > gpurun_out/r06h_bench_semaphore20p_shard1024.json 2> gpurun_out/r06h_bench_semaphore20p_shard1024.err; echo "semaphore rc=$?"
timeout 1200 python bench.py --workload ecdsa_verify > gpurun_out/r06h_bench_ecdsa_verify_1024.json 2> gpurun_out/r06h_bench_ecdsa_verify_1024.err; echo "ecdsa rc=$?"
timeout 600 python bench.py --workload poseidon2_goldilocks > gpurun_out/r06h_bench_poseidon2_goldilocks.json 2> gpurun_out/r06h_bench_poseidon2_goldilocks.err; echo "goldilocks rc=$?"
CW_ARTEFACT_FP=r06b timeout 1500 python bench.py --workload sha256_27008 --batch 262144 --in-flight 4 --steps 8 --no-cpu-baseline > gpurun_out/r06h_bench_sha256_27008_4x2e18.json 2> gpurun_out/r06h_bench_sha256_27008_4x2e18.err; echo "27008 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06h_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("r06h_bench_")[1], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), "roofline", r.get("kernel"), r.get("frac"), "valu", (d.get("roofline_valu") or {}).get("frac"),
              "step", {k: round(d["step"][k], 3) for k in ("input_frac", "all_traffic_frac")}, "parity", (d.get("parity") or {}).get("oracle", "")[:40], "failed", d["failed_instances"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
