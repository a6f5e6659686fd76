"""Throughput of small Semaphore shards with several batches in flight, per strand count:
python tools/sema_inflight.py [wide]   (1 024 instances per batch, own HIP stream per batch; wide = 64 / 32 lanes per
workgroup with 8 / 16 / 32 batches in flight: the throughput end of the latency / throughput trade-off)"""
import os
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from circom_amd import runtime as rt
from circom_amd.compiler import compile_program

name, B = "semaphore20p", 1024
d = tempfile.mkdtemp()
cp = compile_program(bench.make_program(name), d, name, sym=False, strands=(2, 4, 8, 16))
c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
h = bench.synth_inputs(name, c.q, B, c.n_inputs, 3)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(h).to(dev)
for S in ((16,) if len(sys.argv) > 1 else (16, 8, 4, 2)):
    for lanes in ((64, 32) if len(sys.argv) > 1 else (16, 32)):
        for nfl in ((8, 16, 32) if len(sys.argv) > 1 else (4, 8, 16)):
            os.environ["CW_STRANDS"] = str(S)
            os.environ["CW_LANES"] = str(lanes)
            streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
            bs = [c.batch(B, device=0, stream=s.cuda_stream) for s in streams]
            for b in bs:
                b.set_inputs_device(d_in.data_ptr())
            for b in bs:
                b.run(); b.check_r1cs()
            torch.cuda.synchronize()
            steps = 2 * nfl
            t0 = time.perf_counter()
            for k in range(steps):
                bs[k % nfl].run(); bs[k % nfl].check_r1cs()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert (bs[0].status() == 0).all()
            print("S %2d lanes %2d in flight %2d: %8.1f K witnesses/s  (%.2f ms per step)" % (bs[0].strands, bs[0].lanes, nfl, B * steps / dt / 1e3, dt / steps * 1e3), flush=True)
            for b in bs:
                b.close()
