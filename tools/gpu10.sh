set -x
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
python - <<'PY'
import sys, time, tempfile
sys.path.insert(0,'.')
import numpy as np
from bench import build_workload, synth_inputs
from circom_amd import runtime as rt
d=tempfile.mkdtemp()
cp=build_workload("poseidon2", d)
c=rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
B=65536
h=synth_inputs("poseidon2", c.q, B, c.n_inputs, 1)
b=c.batch(B)
for _ in range(2):
    b.set_inputs(h); b.run(); b.check_r1cs(); b.sync()
t=time.perf_counter()
for _ in range(5):
    b.set_inputs(h); b.run(); b.check_r1cs(); b.sync()
dt=(time.perf_counter()-t)/5
print("PCIe-inclusive (host inputs each step): %.3f ms/step -> %.3g witnesses/s"%(dt*1e3, B/dt))
t=time.perf_counter(); w=b.witness_bytes(0); print("egress one witness: %.3f ms"%((time.perf_counter()-t)*1e3))
PY
