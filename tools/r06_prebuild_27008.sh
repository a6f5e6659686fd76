#!/bin/bash
# Lower the 53-block SHA-256 (sha256_27008: 10.8 M constraints at --O0, 1.07 M at the reference's default --O1) for the emitted
# engine - 30 minutes, 17 GB of memory - and leave the artefacts xz-compressed (93 MB) in gpurun_in/cache (they travel with the gpurun
# snapshot; bench.get_compiled unpacks them on first use).  The key r06b names the artefacts whatever the tree's fingerprint is.
#   bash tools/r06_prebuild_27008.sh && CW_ARTEFACT_FP=r06b python bench.py --workload sha256_27008
set -e
cd "$(dirname "$0")/.."
C=${CW_CACHE_27008:-/tmp/cw_cache_27008}
CW_ARTEFACT_FP=r06b python -c "
import bench, os
cp, _, cached = bench.get_compiled('sha256_27008', 1 << 19, '$C', 0, None)
print(cp.dir, cached)
"
D=$C/sha256_27008_s1_b1_ma_r06b
O=gpurun_in/cache/sha256_27008_s1_b1_ma_r06b
mkdir -p $O
for e in cwt dat r1cs; do [ -f $D/sha256_27008.$e.xz ] || xz -T8 -3 -k $D/sha256_27008.$e; cp $D/sha256_27008.$e.xz $O/; done
cp $D/sha256_27008.jit.json $D/sha256_27008.fpjit.json $D/done $O/
echo "prebuilt under the explicit key r06b: __graft_entry__.build() keeps this directory" > $O/keep
du -sh $O
