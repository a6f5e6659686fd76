# A/B of the emitted 256-bit code on a GPU box: bench lines with the emitted code (CW_FP_JIT=1) and with the interpreter
set -x
mkdir -p gpurun_out
for jit in ${JITS:-1}; do
  CW_FP_JIT=$jit python bench.py --workload poseidon2 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_poseidon2_jit$jit.json 2> gpurun_out/ab_poseidon2_jit$jit.err
  CW_FP_JIT=$jit python bench.py --workload semaphore20p --total-batch 8192 --shard-of 8 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ab_sem_shard_jit$jit.json 2> gpurun_out/ab_sem_shard_jit$jit.err
  CW_FP_JIT=$jit python bench.py --workload semaphore20p --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ab_sem8192_jit$jit.json 2> gpurun_out/ab_sem8192_jit$jit.err
done
tail -c 300 gpurun_out/*.err
