timeout 900 python -m pytest tests/test_eddsa.py tests/test_more_circuits.py -m gpu -x -q 2>&1 | tail -5
run() { timeout 600 python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RES $LABEL value %.4g w/s eval %.3f ms r1cs %.3f ms bad %d strands %d'%(d['value'], d['roofline']['kernel_ms'], d['r1cs_check_ms'], d['failed_instances'], d['roofline']['strands']))"; }
LABEL="semaphore20 B=8192" run --workload semaphore20 --batch 8192
LABEL="semaphore20 B=8192 S=4" CW_STRANDS=4 run --workload semaphore20 --batch 8192
LABEL="semaphore20 B=65536" run --workload semaphore20 --batch 65536
LABEL="semaphore20 B=65536 S=1" CW_STRANDS=1 run --workload semaphore20 --batch 65536
LABEL="semaphore20 B=262144 S=1" CW_STRANDS=1 run --workload semaphore20 --batch 262144
