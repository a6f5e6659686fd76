set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu2.log; tail -15 gpurun_out/pytest_gpu2.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench2.log 2>&1; tail -1 gpurun_out/bench2.log
python - <<'PY' 2>&1 | tail -5
import numpy as np, sys
sys.path.insert(0,'.')
from circom_amd import runtime as rt
from circom_amd.field import PRIMES
q=PRIMES['bn128']
rng=np.random.default_rng(5)
n=1<<22
a=rng.integers(0,256,size=(n,32),dtype=np.uint8); a[:,31]&=0x1f
b=rng.integers(0,256,size=(n,32),dtype=np.uint8); b[:,31]&=0x1f
for iters in (256,1024):
    out,ms=rt.fp_mul_bench(q,a,b,iters)
    print("fp mulbench n=%d iters=%d: %.3f ms -> %.2f G mul/s"%(n,iters,ms,n*iters/ms/1e6))
PY
