"""Where does a small Semaphore shard spend its time?  Evaluation time of semaphore20p for a few batch sizes under every
schedule variant (CW_STRANDS / CW_LANES / CW_PIPE are read by cw_batch_create):  python tools/sema_latency.py [levels]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "semaphore20p"
d = tempfile.mkdtemp()
cp = compile_program(bench.make_program(name), d, name, sym=False, strands=(1, 4, 16), pipe=(8, 8))
c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
print("rows", c.n_rows, "mmul", c.n_mmul, "signals", c.n_signals, flush=True)
for B in (64, 1024, 8192):
    h = bench.synth_inputs(name, c.q, B, c.n_inputs, 3)
    for env in ({"CW_STRANDS": "16"}, {"CW_STRANDS": "16", "CW_LANES": "64"}, {"CW_STRANDS": "4"}, {"CW_STRANDS": "4", "CW_LANES": "64"},
                {"CW_STRANDS": "1"}, {"CW_PIPE": "1"}):
        for k in ("CW_STRANDS", "CW_LANES", "CW_PIPE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        b = c.batch(B)
        b.set_inputs(h)
        b.run(); b.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            b.run()
        b.sync()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        assert (b.status() == 0).all()
        print("B %5d %-40s strands %2d lanes %2d pipe %s  eval %8.3f ms" % (B, env, b.strands, b.lanes, b.pipelined, ms), flush=True)
        b.close()
