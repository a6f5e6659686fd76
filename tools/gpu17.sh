timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
run() { timeout 600 python bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RES $LABEL value %.4g w/s eval %.3f ms r1cs %.3f ms bad %d strands %d lanes %d'%(d['value'], d['roofline']['kernel_ms'], d['r1cs_check_ms'], d['failed_instances'], d['roofline']['strands'], d['roofline']['lanes_per_workgroup']))"; }
for L in 64 32 16; do LABEL="sha B=4096 L=$L" CW_LANES=$L run --workload sha256_512 --batch 4096; done
for L in 64 32; do LABEL="sha B=8192 L=$L" CW_LANES=$L run --workload sha256_512 --batch 8192; done
for L in 64 32; do LABEL="semaphore B=8192 L=$L" CW_LANES=$L run --workload semaphore20 --batch 8192; done
LABEL="poseidon default" run
LABEL="poseidon B=16384" run --batch 16384
LABEL="poseidon B=16384 L=64" CW_LANES=64 run --batch 16384
