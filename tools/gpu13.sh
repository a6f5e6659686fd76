timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_more_circuits.py -m gpu -x -q 2>&1 | tail -5
run() { python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RES $LABEL value %.4g w/s eval %.3f ms r1cs %.3f ms bad %d'%(d['value'], d['roofline']['kernel_ms'], d['r1cs_check_ms'], d['failed_instances']))"; }
for PF in 1 2; do for T in 192 768; do LABEL="poseidon stream PF=$PF T=$T" CW_R1CS_PF=$PF CW_R1CS_TERMS=$T run; done; done
for E in 8 10; do LABEL="poseidon staged E=$E CH=12" CW_R1CS_MODE=staged CW_R1CS_ENTRIES=$E CW_R1CS_CHUNKS=12 run; done
for PF in 1 2; do for T in 192 768; do LABEL="sha stream PF=$PF T=$T" CW_R1CS_PF=$PF CW_R1CS_TERMS=$T run --workload sha256_512 --batch 4096; done; done
for E in 8 10; do LABEL="sha staged E=$E CH=192" CW_R1CS_MODE=staged CW_R1CS_ENTRIES=$E CW_R1CS_CHUNKS=192 run --workload sha256_512 --batch 4096; done
