# Per-opcode time breakdown of the schedule evaluation (debug build with -DCW_PROFILE, not the product library).
# usage (CPU box): bash tools/profile_ops.sh build      -> gpurun_in/libcircom_amd_prof.so
#       (GPU box): CW_LIB=gpurun_in/libcircom_amd_prof.so python tools/tape_bench.py <dir> <name> <batch>
set -e
cd "$(dirname "$0")/../circom_amd/csrc"
mkdir -p ../../gpurun_in
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCW_PROFILE -Wno-unused-value -shared -x hip cw_kernels.hip cw_bits.hip cw64.hip cw_host.cpp -o ../../gpurun_in/libcircom_amd_prof.so
