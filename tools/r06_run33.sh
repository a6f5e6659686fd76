#!/bin/bash
# round 6, thirty-third GPU run (final): the driver's three steps on the final library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 1400 python -m pytest tests -x -q -m gpu) > gpurun_out/r06al_gpu_suite.log 2>&1
grep -E "passed|failed" gpurun_out/r06al_gpu_suite.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06al_smoke.log 2>&1; tail -1 gpurun_out/r06al_smoke.log
(time python bench.py --gpus 1 --steps 5 --warmup 1) > gpurun_out/r06al_bench_sha256_2048_2M.json 2> gpurun_out/r06al_default.err; tail -3 gpurun_out/r06al_default.err
tail -1 gpurun_out/r06al_bench_sha256_2048_2M.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.5g' % d['value'], d['ms_per_step'], d['roofline']['frac'], d['step']['all_traffic_frac'], (d.get('parity') or {}).get('parity_checked'), d['cpu_baseline']['value'], d['roofline_fpmul']['frac'])"
