#!/bin/bash
# round 6, twenty-fifth GPU run: the whole suite with -v (which worker ran what before the graph-replay test) and the test's own diagnosis
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1400 python -m pytest tests -m gpu -v > gpurun_out/r06ad_gpu_suite_v.log 2>&1
grep -E "passed|failed" gpurun_out/r06ad_gpu_suite_v.log | tail -1
grep -E "AssertionError: round" gpurun_out/r06ad_gpu_suite_v.log | cut -c1-500
W=$(grep "test_graph_replay_of_emitted_256_bit_code" gpurun_out/r06ad_gpu_suite_v.log | grep -oE "\[gw[0-9]\]" | head -1)
echo "worker $W"
grep -F "$W" gpurun_out/r06ad_gpu_suite_v.log | grep -B40 "test_graph_replay_of_emitted_256_bit_code" | cut -c1-160 | tail -45
