# scheduler cost parameters (CW_ROW_OVERHEAD : CW_EXTRA_COST : CW_AFFINITY_SLACK at lowering time) under the emitted code: the
# Semaphore-style shard and the 8 192 batch, rows alone
for d in gpurun_in/cache_sch_*; do
  for args in "--total-batch 8192 --shard-of 8" ""; do
  CW_FP_FUSED=0 python bench.py --workload semaphore20p $args --steps 6 --warmup 2 --no-cpu-baseline --no-parity --in-flight 1 --cache-dir $d 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$d', '$args', d['isolated'])"
  done
done
