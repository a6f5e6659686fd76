#!/bin/bash
# round 6, sixteenth GPU run: the driver's own command on the tree of this session (default line), the RCCL path on one GPU, and
# the lines of the other BASELINE configs with this session's policies
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time python bench.py) > gpurun_out/r06v_bench_sha256_2048_2M.json 2> gpurun_out/r06v_default.err; tail -3 gpurun_out/r06v_default.err
CW_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06v_bench_rccl_path_one_gpu.json 2> gpurun_out/r06v_rccl.err; tail -2 gpurun_out/r06v_rccl.err
timeout 900 python bench.py --workload sha256_512 --batch 4096 --steps 512 --warmup 64 > gpurun_out/r06v_bench_sha256_512_4096.json 2> gpurun_out/r06v_sha512.err
timeout 900 python bench.py --workload poseidon2 --steps 256 --warmup 16 > gpurun_out/r06v_bench_poseidon2.json 2> gpurun_out/r06v_poseidon2.err
for f in gpurun_out/r06v_bench_*.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', '%.5g' % d['value'], d['ms_per_step'], d['config'].get('in_flight'), d['config'].get('step_launch', '')[:10], d['isolated'].get('kernels_ms'), d['roofline'].get('frac'), (d.get('parity') or {}).get('parity_checked'), (d.get('cpu_baseline') or {}).get('value'), d.get('step', {}).get('all_traffic_frac'))"; done
