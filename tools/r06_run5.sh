#!/bin/bash
# round 6, fifth GPU run: what the closing run left open - the whole GPU suite again (audit-spill test with a starved audit, the
# 53-block SHA-256 from its xz artefacts), the Goldilocks engine's new kernels (tests + line), sha256_27008 with four batches of
# 2^18 in flight, the RCCL path on one GPU, and the default line once more (now quoting the r06h counters)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06i_gpu_suite.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/r06i_gpu_suite.log
timeout 600 python bench.py --workload poseidon2_goldilocks > gpurun_out/r06i_bench_poseidon2_goldilocks.json 2> gpurun_out/r06i_bench_poseidon2_goldilocks.err; echo "goldilocks rc=$?"
CW_ARTEFACT_FP=r06b timeout 1500 python bench.py --workload sha256_27008 --batch 262144 --in-flight 4 --steps 8 --no-cpu-baseline > gpurun_out/r06i_bench_sha256_27008_4x2e18.json 2> gpurun_out/r06i_bench_sha256_27008_4x2e18.err; echo "27008 rc=$?"
CW_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 900 python bench.py --steps 5 --no-cpu-baseline --no-small > gpurun_out/r06i_bench_rccl_path_one_gpu.json 2> gpurun_out/r06i_bench_rccl_path_one_gpu.err; echo "rccl rc=$?"
timeout 900 python bench.py > gpurun_out/r06i_bench_sha256_2048_2M.json 2> gpurun_out/r06i_bench_sha256_2048_2M.err; echo "default rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06i_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("r06i_bench_")[1], "value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), "roofline", r.get("kernel"), r.get("frac"), "traffic", r.get("traffic"),
              "kernels", d.get("isolated", {}).get("kernels_ms"), "parity", (d.get("parity") or {}).get("oracle", "")[:50], "failed", d["failed_instances"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
