"""Two batches in flight on two streams (does the R1CS check of one overlap the evaluation of the other?):
python tools/tape_bench2.py <dir> <name> <batch> [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from circom_amd import runtime as rt
d, name, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
c = rt.Circuit(os.path.join(d, name + ".cwt"), os.path.join(d, name + ".dat"), os.path.join(d, name + ".r1cs"))
rng = np.random.default_rng(1)
arr = rng.integers(0, 256, size=(B, c.n_inputs, 32), dtype=np.uint8); arr[:, :, 31] &= 0x0F
din = torch.from_numpy(arr).cuda()
for nfl in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    bs = [c.batch(B, device=0, stream=s.cuda_stream) for s in streams]
    for b in bs:
        b.set_inputs_device(din.data_ptr()); b.run(); b.check_r1cs()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        b = bs[k % nfl]
        b.run(); b.check_r1cs()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all((b.status() == 0).all() for b in bs)
    print("TB2 %s %s B=%d S=%d inflight=%d: %.3f ms/step  %.4g w/s ok=%s" % (d, name, B, bs[0].strands, nfl, dt / steps * 1e3, B * steps / dt, ok))
    for b in bs: b.close()
