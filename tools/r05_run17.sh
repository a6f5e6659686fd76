# round 5, GPU call 17 (closing): the emitted kernel with 4-byte VOP2 forms for gates with a constant operand; ingest / egress as in calls 13 / 15:
# the whole GPU suite, smoke, the default line, rocprofv3 trace + PMC passes of the default command
set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
(time timeout 1400 python -m pytest tests -m gpu -q --durations=12) > gpurun_out/r05r_gpu_suite.log 2>&1
tail -20 gpurun_out/r05r_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05r_smoke.log 2>&1; tail -2 gpurun_out/r05r_smoke.log
(time python bench.py) > gpurun_out/r05r_bench_sha256_2048_2M.json 2> gpurun_out/r05r_default.err; tail -4 gpurun_out/r05r_default.err
rm -f gpurun_out/traffic.json
bash tools/profile.sh r05r_sha256_2048_2M sha256_2048:2097152 2>&1 | tail -30
for f in gpurun_out/r05r_bench_*.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '%.5g' % d['value'], d['ms_per_step'], d['isolated'].get('kernels_ms'), d.get('in_step_kernels_ms'), d['roofline'].get('frac'), d['roofline'].get('traffic'), (d.get('parity') or {}).get('parity_checked'), (d.get('cpu_baseline') or {}).get('value'), d['config'].get('compile_cached'), {k: v for k, v in (d.get('value_canonical_O1') or {}).items() if k in ('witnesses_per_s', 'GB/s', 'chunk_instances', 'egress_ms', 'error')}, (d.get('canonical_egress') or {}).get('GB/s'))"; done
