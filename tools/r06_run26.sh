#!/bin/bash
# round 6, twenty-sixth GPU run: the captured paths launch kernels only (the finding words are reset by a fill kernel, not a memset node):
# the whole GPU suite twice + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in 1 2; do
  (time timeout 1400 python -m pytest tests -m gpu -q --durations=4) > gpurun_out/r06ae_gpu_suite_$k.log 2>&1
  grep -E "passed|failed|FAILED|AssertionError: round" gpurun_out/r06ae_gpu_suite_$k.log | cut -c1-400
done
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06ae_smoke.log 2>&1; tail -1 gpurun_out/r06ae_smoke.log
