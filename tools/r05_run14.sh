# round 5, GPU call 14: cache policy of the emitted kernel's row stores (streaming `nt` as shipped / default), kernel alone, packed inputs
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out gpurun_in/jit
C=gpurun_in/cache
gunzip -c $C/sha256_2048_s1_b1_ma_f29c3856c7801468/sha256_2048.cwt.gz > gpurun_in/jit/sha256_2048.cwt
gunzip -c $C/sha256_2048_s1_b1_ma_f29c3856c7801468/sha256_2048.r1cs.gz > gpurun_in/jit/sha256_2048.r1cs
cp $C/sha256_2048_s1_b1_ma_f29c3856c7801468/sha256_2048.dat gpurun_in/jit/
gunzip -c $C/sha256_2048_s1_b1_ma_expplain/sha256_2048.cwt.gz > gpurun_in/jit/sha256_2048_plain.cwt
for TAG in "" plain "" plain; do
  ENGINES=jit NO_AUDIT=1 TAG=$TAG timeout 300 python tools/jit_bench.py 2048 2097152 6 2>&1 | tail -1 | cut -c1-400
done > gpurun_out/r05n_jit_store_policy.txt 2>&1
cat gpurun_out/r05n_jit_store_policy.txt
