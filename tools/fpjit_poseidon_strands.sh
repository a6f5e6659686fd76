# Poseidon(2) x 65 536 through the emitted code with 1, 2 and 4 strands, rows alone and with the fused check
for S in 1 2 4; do for fused in 0 1; do
  CW_BENCH_STRANDS=$S CW_STRANDS=$S CW_FP_FUSED=$fused python bench.py --workload poseidon2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('S=$S fused=$fused', round(d['value']), {k: round(v, 3) for k, v in d['isolated'].items()}, d['roofline_eval'].get('strands'), (d.get('parity') or {}).get('parity_checked'))"
done; done
