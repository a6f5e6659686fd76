#!/bin/bash
# round 6, twenty-fourth GPU run: the graph-replay failure that only shows in the whole suite - the same tests in ONE process behind the
# files that precede them there
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for pre in tests/test_gpu_parity.py tests/test_fpjit.py tests/test_functions.py "tests/test_montgomery.py tests/test_opzoo.py"; do
  echo "== behind $pre"
  timeout 900 python -m pytest $pre tests/test_run_check_graph.py -q -m gpu -n 0 -p no:cacheprovider 2>&1 | grep -E "passed|failed|AssertionError: round" | cut -c1-400
done 2>&1 | tee gpurun_out/r06ac_graph_debug.log
