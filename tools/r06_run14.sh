#!/bin/bash
# round 6, fourteenth GPU run: how many Semaphore-shard evaluations really run side by side (rocprofv3 kernel trace of the
# 32-in-flight line), and what ONE launch of 16 384 instances with full waves takes (256 workgroups, nothing beside it)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out/prof_r06t
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r06t/trace -- python $R/bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 192 --warmup 32 --no-cpu-baseline --no-parity --graph off > $R/gpurun_out/prof_r06t/trace.log 2>&1
tail -1 $R/gpurun_out/prof_r06t/trace.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('traced', '%.5g' % d['value'], d['ms_per_step'], d['config'].get('in_flight'), d['config'].get('lanes_per_wave'))"
f=$(find $R/gpurun_out/prof_r06t/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/kernel_overlap.py $f | tee $R/gpurun_out/r06t_sema_shard_overlap.txt
find $R/gpurun_out/prof_r06t -name "*.csv" -size +4M -delete
cd $R
for spec in "16384 64 1" "16384 64 2" "4096 64 8" "4096 64 4" "2048 64 16"; do
  set -- $spec
  CW_LANES=$2 timeout 600 python bench.py --workload semaphore20p --batch $1 --in-flight $3 --steps 48 --warmup 8 --no-cpu-baseline --no-parity --graph off > gpurun_out/r06t_sema_b$1_l$2_if$3.json 2> gpurun_out/r06t_sema_b$1_l$2_if$3.err
  tail -1 gpurun_out/r06t_sema_b$1_l$2_if$3.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch $1 lanes $2 in flight $3:', '%.5g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['isolated']['kernels_ms'], d['config'].get('engine'))"
done
