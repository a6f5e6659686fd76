# Timing-only variants of the emitted code (CW_FPJIT_EXP at lowering time: results are garbage, only the clock is read):
# what a launch of the Semaphore-style shard spends on operand waits, barriers, stores and the bodies' arithmetic
for e in nowait nobarrier nostore nocall; do
  python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --in-flight 1 --cache-dir gpurun_in/cache_$e 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e', d['isolated'])"
done
