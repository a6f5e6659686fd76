set -x
mkdir -p gpurun_out
nproc; free -g | head -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_gpu.log; tail -25 gpurun_out/pytest_gpu.log
./tools/ubench_valu > gpurun_out/ubench.log 2>&1; cat gpurun_out/ubench.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench1.log 2>&1; tail -3 gpurun_out/bench1.log
