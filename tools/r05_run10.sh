# round 5, GPU call 10: the boolean-row path of the R1CS stream check (first-bad-row tests, ECDSA goldens, check time A/B)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ecdsa.py tests/test_baseline_configs.py tests/test_eddsa.py tests/test_more_circuits.py tests/test_opzoo.py tests/test_status_word.py tests/test_montgomery.py -m gpu -q > gpurun_out/r05j_tests.log 2>&1
tail -5 gpurun_out/r05j_tests.log
for v in on off on off; do
  if [ "$v" = "on" ]; then unset CW_R1CS_NO_BOOL; else export CW_R1CS_NO_BOOL=1; fi
  timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline --in-flight 1 > gpurun_out/r05j_ecdsa_bool_$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r05j_ecdsa_bool_$v.json').read().strip().splitlines()[-1]); print('bool rows $v', '%.5g' % d['value'], d['isolated']['kernels_ms'], d['parity_checked'])"
done
unset CW_R1CS_NO_BOOL
timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05j_bench_ecdsa_verify_1024.json 2>/dev/null
timeout 600 python bench.py --workload semaphore20p --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r05j_bench_semaphore20p_8192.json 2>/dev/null
for f in gpurun_out/r05j_bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', '%.5g' % d['value'], d['isolated']['kernels_ms'], d['parity_checked'])"; done
