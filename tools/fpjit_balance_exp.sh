# strand load balance of the level scheduler (CW_BALANCE at lowering time: -1 = rounds 1-3 (ties to the first strand), 0 = ties
# to the strand with the least work so far, > 0 = affinity also yields above that multiple of the average load)
for d in gpurun_in/cache_bal_*; do
  for args in "--total-batch 8192 --shard-of 8" ""; do
  CW_FP_FUSED=0 python bench.py --workload semaphore20p $args --steps 6 --warmup 2 --no-cpu-baseline --in-flight 1 --cache-dir $d 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$d', '$args', d['isolated'], (d.get('parity') or {}).get('parity_checked'))"
  done
done
