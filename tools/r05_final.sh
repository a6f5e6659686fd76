# Round-5 evidence on one MI355X (gpurun box): the whole GPU suite, smoke, the default line, rocprofv3 (kernel trace + PMC passes) of the
# default command and the kernel trace of config 5, and a bench line of every BASELINE config, into gpurun_out/r05z_*
set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
(time timeout 1400 python -m pytest tests -m gpu -q --durations=25) > gpurun_out/r05z_gpu_suite.log 2>&1
tail -32 gpurun_out/r05z_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05z_smoke.log 2>&1; tail -2 gpurun_out/r05z_smoke.log
python bench.py > gpurun_out/r05z_bench_sha256_2048_2M.json 2> gpurun_out/r05z_default.err
rm -f gpurun_out/traffic.json
bash tools/profile.sh r05z_sha256_2048_2M sha256_2048:2097152 2>&1 | tail -40
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05z_ecdsa/trace -- python $R/bench.py --workload ecdsa_verify --steps 2 --warmup 1 --no-cpu-baseline --no-parity --fp-bench-lanes 65536 > $R/gpurun_out/prof_r05z_ecdsa/trace.log 2>&1
cd $R
python tools/summarize_prof.py gpurun_out/prof_r05z_ecdsa > gpurun_out/prof_r05z_ecdsa/summary.txt 2>&1; head -12 gpurun_out/prof_r05z_ecdsa/summary.txt
find gpurun_out/prof_r05z_ecdsa -name "*.csv" -size +4M -delete
python bench.py --workload ecdsa_verify --steps 3 --warmup 1 > gpurun_out/r05z_bench_ecdsa_verify_1024.json 2>/dev/null
python bench.py --workload ecdsa_verify --total-batch 1024 --shard-of 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05z_bench_ecdsa_verify_shard128.json 2>/dev/null
python bench.py --workload poseidon2 --steps 20 --warmup 3 > gpurun_out/r05z_bench_poseidon2.json 2>/dev/null
python bench.py --workload poseidon2_goldilocks --steps 20 --warmup 3 > gpurun_out/r05z_bench_poseidon2_goldilocks.json 2>/dev/null
python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 10 --warmup 2 > gpurun_out/r05z_bench_semaphore20p_shard1024.json 2>/dev/null
python bench.py --workload semaphore20p --steps 10 --warmup 2 > gpurun_out/r05z_bench_semaphore20p_8192.json 2>/dev/null
python bench.py --workload semaphore20w --total-batch 8192 --shard-of 8 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r05z_bench_semaphore20w_shard1024.json 2>/dev/null
python bench.py --workload bigmultmodp --steps 10 --warmup 2 > gpurun_out/r05z_bench_bigmultmodp_8192.json 2>/dev/null
python bench.py --workload sha256_512 --batch 4096 --steps 20 --warmup 3 > gpurun_out/r05z_bench_sha256_512_4096.json 2>/dev/null
for f in gpurun_out/r05z_bench_*.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '%.5g' % d['value'], d['isolated'].get('kernels_ms'), (d.get('parity') or {}).get('parity_checked'), (d.get('cpu_baseline') or {}).get('value'))"; done
