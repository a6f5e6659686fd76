"""Two batches on two streams, alternating: does the R1CS check of one batch overlap the evaluation of the other?
python tools/overlap_bench.py <dir> <name> <batch> [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from circom_amd import runtime as rt
d, name, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
c = rt.Circuit(os.path.join(d, name + ".cwt"), os.path.join(d, name + ".dat"), os.path.join(d, name + ".r1cs"))
rng = np.random.default_rng(1)
if name.startswith("sha"):
    arr = np.zeros((B, c.n_inputs, 32), dtype=np.uint8); arr[:, :, 0] = rng.integers(0, 2, size=(B, c.n_inputs), dtype=np.uint8)
else:
    if name.startswith("semaphore"):
        import random
        from circom_amd.circuits import eddsa_host as H
        r = random.Random(1)
        pool = [H.semaphore_inputs(c.q, 20, r)[0] for _ in range(32)]
        one = np.frombuffer(b"".join(v.to_bytes(32, "little") for row in pool for v in row), dtype=np.uint8).reshape(len(pool), c.n_inputs, 32)
        arr = np.ascontiguousarray(np.tile(one, ((B + 31) // 32, 1, 1))[:B])
    else:
        arr = rng.integers(0, 256, size=(B, c.n_inputs, 32), dtype=np.uint8); arr[:, :, 31] &= 0x0F
din = torch.from_numpy(arr).cuda()
nfl = int(os.environ.get("OV_N", "2"))
streams = [torch.cuda.Stream() for _ in range(nfl)]
batches = [c.batch(B, device=0, stream=s.cuda_stream) for s in streams]
for b in batches:
    b.set_inputs_device(din.data_ptr()); b.run(); b.check_r1cs()
torch.cuda.synchronize()
def run(n_batches):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        b = batches[i % n_batches]
        b.set_inputs_device(din.data_ptr()); b.run(); b.check_r1cs()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for n in (1, nfl, 1, nfl):
    print("OV %s B=%d batches in flight %d: %.3f ms/step" % (name, B, n, run(n)))
assert all((b.status() == 0).all() for b in batches) or not (name.startswith("sha") or name.startswith("semaphore"))
