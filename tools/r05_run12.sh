# round 5, GPU call 12 (closing): the whole suite + smoke on the final tree, config 5's lines, rocprofv3 (trace + PMC passes) of config 5
set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
(time timeout 1400 python -m pytest tests/ -x -q -m gpu --durations=6) > gpurun_out/r05m_gpu_suite.log 2>&1
tail -10 gpurun_out/r05m_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05m_smoke.log 2>&1; tail -1 gpurun_out/r05m_smoke.log
timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 > gpurun_out/r05m_bench_ecdsa_verify_1024.json 2>/dev/null
timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline --in-flight 1 > gpurun_out/r05m_bench_ecdsa_verify_1024_one_in_flight.json 2>/dev/null
timeout 600 python bench.py --workload ecdsa_verify --total-batch 1024 --shard-of 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05m_bench_ecdsa_verify_shard128_of_1024.json 2>/dev/null
for f in gpurun_out/r05m_bench_*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', '%.5g' % d['value'], d['isolated']['kernels_ms'], d['parity_checked'], (d.get('one_shot_job') or {}).get('predicted_witnesses_per_s'))"; done
bash tools/profile.sh r05m_ecdsa ecdsa_verify:1024 --workload ecdsa_verify 2>&1 | tail -25
(time python bench.py) > gpurun_out/r05m_bench_default.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r05m_bench_default.json').read().strip().splitlines()[-1]); print('default value %.5g' % d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity_checked'])"
