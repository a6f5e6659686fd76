# round 5, GPU call 5: ECDSA verifier with longest-first balancing of interpreted levels; the goldilocks bench line
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
CW_ARTEFACT_FP=r05exp4 timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline --in-flight 1 > gpurun_out/r05e_bench_ecdsa_1024_one_in_flight.json 2> gpurun_out/r05e_bench_ecdsa_1024_one.err
tail -2 gpurun_out/r05e_bench_ecdsa_1024_one.err
CW_ARTEFACT_FP=r05exp4 timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 > gpurun_out/r05e_bench_ecdsa_1024.json 2> gpurun_out/r05e_bench_ecdsa_1024.err
tail -2 gpurun_out/r05e_bench_ecdsa_1024.err
timeout 600 python bench.py --workload poseidon2_goldilocks --steps 20 --warmup 2 > gpurun_out/r05e_bench_poseidon2_goldilocks.json 2> gpurun_out/r05e_bench_poseidon2_goldilocks.err
tail -3 gpurun_out/r05e_bench_poseidon2_goldilocks.err; cut -c1-600 gpurun_out/r05e_bench_poseidon2_goldilocks.json
d=gpurun_in/cache/ecdsa_verify_s16_b1_ma_r05exp4
gunzip -k $d/ecdsa_verify.r1cs.gz
CW_LIB=gpurun_in/libcircom_amd_prof.so timeout 600 python tools/tape_bench.py $d ecdsa_verify 1024 1 > gpurun_out/r05e_ecdsa_prof.log 2>&1
grep "PROF all\|ARRIVE\|^TB" gpurun_out/r05e_ecdsa_prof.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05e_bench_ecdsa*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], "isolated", d["isolated"].get("kernels_ms"), "in_step", d.get("in_step_kernels_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
