# Build timing-experiment variants of the library (never the product build): tools/bits_exp.sh  -> gpurun_in/exp/lib_<name>.so
# NORECS: the bit-plane evaluation kernel fetches no records (every batch replays the first): what the record stream costs.
set -e
cd "$(dirname "$0")/../circom_amd/csrc"
mkdir -p ../../gpurun_in/exp build
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-unused-value -w"
for v in NORECS NOCMD BOTH; do
  D="-DCW_EXP_$v"
  [ $v = BOTH ] && D="-DCW_EXP_NORECS -DCW_EXP_NOCMD"
  /opt/rocm/bin/hipcc $FL $D -x hip -c cw_bits.hip -o build/cw_bits_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/cw_kernels.hip.o build/cw_bits_$v.o build/cw_host.cpp.o -o ../../gpurun_in/exp/lib_$v.so
done
ls -la ../../gpurun_in/exp/
