set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu5.log; tail -6 gpurun_out/pytest_gpu5.log
for S in 1 4 16; do CW_STRANDS=$S python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('poseidon S=$S value %.3g w/s eval %.3f ms r1cs %.3f ms frac %.3f'%(d['value'], d['roofline']['kernel_ms'], d['r1cs_check_ms'], d['roofline']['frac']))"; done
for S in 1 4 16; do CW_STRANDS=$S python bench.py --workload sha256_512 --batch 4096 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sha256_512 S=$S value %.3g w/s eval %.3f ms r1cs %.3f ms frac %.3f bad %d'%(d['value'], d['roofline']['kernel_ms'], d['r1cs_check_ms'], d['roofline']['frac'], d['failed_instances']))"; done
