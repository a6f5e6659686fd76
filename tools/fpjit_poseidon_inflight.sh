# Poseidon(2) x 65 536, single-strand emitted code with the fused check: batches in flight (one wave per SIMD per batch)
for S in 1 4; do for nf in 2 3 4 6; do
  CW_BENCH_STRANDS=$S CW_STRANDS=$S CW_FP_FUSED=1 python bench.py --workload poseidon2 --steps 24 --warmup 6 --no-cpu-baseline --in-flight $nf 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S=$S in_flight=$nf', round(d['value']), round(d['ms_per_step'],3), (d.get('parity') or {}).get('parity_checked'))"
done; done
