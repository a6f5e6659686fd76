#!/bin/bash
# round 6, third GPU run: where do the emitted SHA kernel's 12-13 ms go?  Timing-only variants (tools/prebuild_jit_variants.py:
# wrong results by construction) of the straight-line code of sha256_2048 at 2^21 instances, packed inputs (the evaluation alone)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
gunzip gpurun_in/jit/*.gz
for tag in ${TAGS:-base}; do
  NO_AUDIT=1 TAG=$tag ENGINES=jit timeout 300 python tools/jit_bench.py 2048 2097152 6 > gpurun_out/r06${RUNID:-c}_jit_$tag.json 2> gpurun_out/r06${RUNID:-c}_jit_$tag.err
  echo "$tag rc=$? $(tail -c 600 gpurun_out/r06${RUNID:-c}_jit_$tag.json)"
done
