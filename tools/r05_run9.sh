# round 5, GPU call 9: wave priority by age (ECDSA, A/B), BigMultModP with the single-strand variant back, the repaired tests, rocprofv3 of config 5
set -x
export TMPDIR=/tmp CW_ARTEFACT_FP=2311f4478d3a2fc5
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ecdsa.py tests/test_functions.py tests/test_fpjit.py -m gpu -q > gpurun_out/r05i_tests.log 2>&1
tail -5 gpurun_out/r05i_tests.log
for m in 0 0x80000000 0 0x80000000; do
  if [ "$m" = "0" ]; then unset CW_PRIO_MASK; else export CW_PRIO_MASK=$m; fi
  timeout 600 python bench.py --workload ecdsa_verify --steps 3 --warmup 1 --no-cpu-baseline --no-parity --in-flight 1 > gpurun_out/r05i_ecdsa_prio_$m.json 2> gpurun_out/r05i_ecdsa_prio.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05i_ecdsa_prio_$m.json").read().strip().splitlines()[-1])
print("CW_PRIO_MASK=$m", "value %.5g" % d["value"], d["isolated"]["kernels_ms"])
PY
done
unset CW_PRIO_MASK CW_ARTEFACT_FP
for B in 8192 65536; do
  timeout 600 python bench.py --workload bigmultmodp --batch $B --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r05i_bench_bigmultmodp_$B.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r05i_bench_bigmultmodp_$B.json').read().strip().splitlines()[-1]); print('bigmultmodp $B', '%.5g' % d['value'], d['roofline']['strands'], d['roofline']['kernel'], d['isolated']['kernels_ms'])"
done
export CW_ARTEFACT_FP=2311f4478d3a2fc5
mkdir -p gpurun_out/prof_r05i_ecdsa
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05i_ecdsa/trace -- python $R/bench.py --workload ecdsa_verify --steps 2 --warmup 1 --no-cpu-baseline --no-parity --fp-bench-lanes 65536 > $R/gpurun_out/prof_r05i_ecdsa/trace.log 2>&1
cd $R
python tools/summarize_prof.py gpurun_out/prof_r05i_ecdsa > gpurun_out/prof_r05i_ecdsa/summary.txt 2>&1; head -8 gpurun_out/prof_r05i_ecdsa/summary.txt
find gpurun_out/prof_r05i_ecdsa -name "*.csv" -size +4M -delete
