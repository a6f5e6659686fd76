# round 5, GPU call 11: the whole suite on the tree with the boolean-row check; four operand loads in flight for sums (ECDSA, A/B)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
(time timeout 1400 python -m pytest tests/ -x -q -m gpu --durations=6) > gpurun_out/r05k_gpu_suite.log 2>&1
tail -12 gpurun_out/r05k_gpu_suite.log
for v in 0 1 0 1; do
  CW_WIDE_LINSUM=$v timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline --no-parity --in-flight 1 > gpurun_out/r05k_ecdsa_wide$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r05k_ecdsa_wide$v.json').read().strip().splitlines()[-1]); print('wide linsum $v', '%.5g' % d['value'], d['isolated']['kernels_ms'])"
done
