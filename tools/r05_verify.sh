# the driver's three steps on the final tree: the whole GPU suite, smoke(), the default bench line
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
(time timeout 1400 python -m pytest tests/ -x -q -m gpu --durations=12) > gpurun_out/r05y_gpu_suite.log 2>&1
tail -20 gpurun_out/r05y_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05y_smoke.log 2>&1; tail -2 gpurun_out/r05y_smoke.log
(time python bench.py) > gpurun_out/r05y_bench_default.json 2> gpurun_out/r05y_bench_default.err; tail -4 gpurun_out/r05y_bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r05y_bench_default.json').read().strip().splitlines()[-1]); print('value %.5g' % d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['compile_cached'], d['parity_checked'], d['roofline'].get('traffic'), (d['cpu_baseline'] or {}).get('value'), (d.get('value_canonical_O1') or {}).get('witnesses_per_s'))"
