for S in 16 8 4; do
  CW_BENCH_STRANDS=$S CW_STRANDS=$S python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --in-flight 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S=$S', 'shard1024', d['isolated'], d['roofline_eval'].get('strands'), d['roofline_eval'].get('lanes_per_workgroup'))"
  CW_BENCH_STRANDS=$S CW_STRANDS=$S python bench.py --workload semaphore20p --steps 6 --warmup 2 --no-cpu-baseline --no-parity --in-flight 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S=$S', 'x8192', d['isolated'], d['roofline_eval'].get('strands'), d['roofline_eval'].get('lanes_per_workgroup'))"
done
