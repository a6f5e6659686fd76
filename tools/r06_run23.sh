#!/bin/bash
# round 6, twenty-third GPU run: the graph-replay tests six times over (the closing run had ONE failure of
# test_graph_replay_of_emitted_256_bit_code: the emitted kernels' argument blocks were locals of cw_run - now members of the batch),
# then the whole GPU suite and smoke on that tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_run_check_graph.py -q -m gpu -n 4 2>&1 | tail -1
done | tee gpurun_out/r06ab_graph_tests_x6.log
(time timeout 1400 python -m pytest tests -m gpu -q --durations=4) > gpurun_out/r06ab_gpu_suite.log 2>&1
tail -4 gpurun_out/r06ab_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06ab_smoke.log 2>&1; tail -1 gpurun_out/r06ab_smoke.log
