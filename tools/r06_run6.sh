#!/bin/bash
# round 6, sixth GPU run: are the multi-batch lines (configs 2-5) slower than round 5's, or is it the box?  The round-5 tree
# (gpurun_in/r05tree: commit ba8c06f with its own library) and this tree run the same workloads back to back on ONE box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06j_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/r06j_smoke.log)"
timeout 600 python -m pytest tests/test_goldilocks_device.py -m gpu -q -n 2 > gpurun_out/r06j_goldilocks_tests.log 2>&1; echo "goldilocks tests rc=$? $(tail -1 gpurun_out/r06j_goldilocks_tests.log)"
timeout 600 python bench.py --workload poseidon2_goldilocks > gpurun_out/r06j_bench_poseidon2_goldilocks.json 2> gpurun_out/r06j_bench_poseidon2_goldilocks.err
python -c "
import json
d=json.loads(open('gpurun_out/r06j_bench_poseidon2_goldilocks.json').read().strip().splitlines()[-1]); print('goldilocks value %.4g ms/step %.3f' % (d['value'], d['ms_per_step']), d['isolated'].get('kernels_ms'), d['parity'])"
for rep in 1 2; do
  for w in "sha256_512 --batch 4096" "poseidon2" "semaphore20p --total-batch 8192 --shard-of 8"; do
    tag=$(echo $w | cut -d' ' -f1)
    for tree in new old; do
      if [ $tree = new ]; then B=bench.py; else B=gpurun_in/r05tree/bench.py; fi
      timeout 600 python $B --workload $w --no-parity --no-cpu-baseline --steps 20 > gpurun_out/r06j_${tag}_${tree}_$rep.json 2> gpurun_out/r06j_${tag}_${tree}_$rep.err
      python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r06j_${tag}_${tree}_$rep.json").read().strip().splitlines()[-1])
    print("$tag $tree $rep value %.4g ms/step %.4f in_flight %s" % (d["value"], d["ms_per_step"], d["config"]["in_flight"]), d["isolated"].get("kernels_ms"))
except Exception as e:
    print("$tag $tree $rep unreadable", e)
PY
    done
  done
done
