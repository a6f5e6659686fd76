# round 5, GPU call 16: gates with a constant operand as 4-byte VOP2 instructions (-13 % code) against the all-bitop3 code, kernel alone
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out gpurun_in/jit
C=gpurun_in/cache/sha256_2048_s1_b1_ma_72a127233ffeada5
gunzip -c $C/sha256_2048.cwt.gz > gpurun_in/jit/sha256_2048.cwt
gunzip -c $C/sha256_2048.r1cs.gz > gpurun_in/jit/sha256_2048.r1cs
cp $C/sha256_2048.dat gpurun_in/jit/
gunzip -c gpurun_in/jit_old/sha256_2048_old.cwt.gz > gpurun_in/jit/sha256_2048_old.cwt
for TAG in old "" old ""; do
  ENGINES=jit NO_AUDIT=1 TAG=$TAG timeout 300 python tools/jit_bench.py 2048 2097152 6 2>&1 | tail -1 | cut -c1-330
done > gpurun_out/r05q_jit_vop2.txt 2>&1
cat gpurun_out/r05q_jit_vop2.txt
