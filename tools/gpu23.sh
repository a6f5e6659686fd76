timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_more_circuits.py tests/test_eddsa.py tests/test_golden_wtns.py -m gpu -x -q 2>&1 | tail -4
run() { timeout 600 python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RES $LABEL value %.4g w/s eval %.3f ms r1cs %.3f ms bad %d'%(d['value'], d['roofline']['kernel_ms'], d['r1cs_check_ms'], d['failed_instances']))"; }
LABEL="poseidon" run
LABEL="semaphore 8192" run --workload semaphore20 --batch 8192
