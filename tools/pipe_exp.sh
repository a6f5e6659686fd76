# Build timing-experiment variants of the library (never the product build): tools/pipe_exp.sh -> gpurun_in/exp/lib_p<name>.so
set -e
cd "$(dirname "$0")/../circom_amd/csrc"
mkdir -p ../../gpurun_in/exp
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-unused-value -w"
for v in BASE NOSTORE NOLOAD "NOSTORE -DCW_PEXP_NOLOAD" $PIPE_EXP_EXTRA; do
  name=$(echo $v | tr -d ' ' | tr -d '-' | sed 's/DCW_PEXP_//g')
  /opt/rocm/bin/hipcc $FL -DCW_PEXP_$v -shared -x hip cw_kernels.hip cw_bits.hip cw_host.cpp -o ../../gpurun_in/exp/lib_p$name.so &
done
wait
ls -la ../../gpurun_in/exp/
