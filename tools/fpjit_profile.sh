# rocprofv3 evidence for the emitted 256-bit code (run on the GPU box): kernel trace + PMC passes of Poseidon(2) x 65 536
# (fused check) and of the Semaphore-style shard (rows alone + stand-alone check), then the bench lines of both
set -x
mkdir -p gpurun_out
rm -f gpurun_out/traffic.json
bash tools/profile.sh r04c_poseidon2 poseidon2:65536 --workload poseidon2 2>&1 | tail -30
bash tools/profile.sh r04c_semaphore20p_shard1024 semaphore20p:1024 --workload semaphore20p --total-batch 8192 --shard-of 8 --in-flight 1 2>&1 | tail -30
