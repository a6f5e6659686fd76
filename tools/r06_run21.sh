#!/bin/bash
# round 6, twenty-first GPU run (closing): the driver's three steps on the final tree + rocprofv3 trace / PMC passes of the default command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1400 python -m pytest tests -m gpu -q --durations=6) > gpurun_out/r06zz_gpu_suite.log 2>&1
tail -4 gpurun_out/r06zz_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06zz_smoke.log 2>&1; tail -1 gpurun_out/r06zz_smoke.log
(time python bench.py) > gpurun_out/r06zz_bench_sha256_2048_2M.json 2> gpurun_out/r06zz_default.err; tail -3 gpurun_out/r06zz_default.err
cp profiles/traffic.json gpurun_out/traffic.json
bash tools/profile.sh r06zz_sha256_2048_2M sha256_2048:2097152 2>&1 | tail -25
timeout 600 python bench.py --workload poseidon2 --steps 300 --warmup 30 > gpurun_out/r06zz_bench_poseidon2.json 2> gpurun_out/r06zz_poseidon2.err
for f in gpurun_out/r06zz_bench_*.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '%.5g' % d['value'], d['ms_per_step'], d['isolated'].get('kernels_ms'), d.get('in_step_kernels_ms'), d['roofline'].get('frac'), (d.get('parity') or {}).get('parity_checked'), (d.get('cpu_baseline') or {}).get('value'), (d.get('roofline_valu') or {}).get('frac'), ((d.get('roofline_valu') or {}).get('incl_check') or {}).get('frac'))"; done
