timeout 900 python -m pytest tests/test_eddsa.py -m gpu -x -q 2>&1 | tail -5
run() { timeout 600 python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RES $LABEL value %.4g w/s eval %.3f ms r1cs %.3f ms bad %d strands %d mmul/w %d'%(d['value'], d['roofline']['kernel_ms'], d['r1cs_check_ms'], d['failed_instances'], d['roofline']['strands'], d['config']['fp_mul_per_witness']))"; }
LABEL="semaphore20 B=1024" run --workload semaphore20 --batch 1024
LABEL="semaphore20 B=8192" run --workload semaphore20 --batch 8192
LABEL="semaphore20 B=8192 S=4" CW_STRANDS=4 run --workload semaphore20 --batch 8192
LABEL="semaphore20 B=65536" run --workload semaphore20 --batch 65536
