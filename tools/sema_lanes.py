import os, sys, tempfile, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, bench
from circom_amd import runtime as rt
from circom_amd.compiler import compile_program
name = "semaphore20p"
d = tempfile.mkdtemp()
cp = compile_program(bench.make_program(name), d, name, sym=False, strands=(4, 16))
c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
for B in (8192, 32768):
    h = bench.synth_inputs(name, c.q, B, c.n_inputs, 3)
    for env in ({"CW_STRANDS": "16", "CW_LANES": "16"}, {"CW_STRANDS": "16", "CW_LANES": "32"}, {"CW_STRANDS": "16", "CW_LANES": "64"}, {"CW_STRANDS": "4", "CW_LANES": "16"}, {"CW_STRANDS": "4", "CW_LANES": "32"}):
        for k in ("CW_STRANDS", "CW_LANES", "CW_PIPE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            b = c.batch(B)
        except Exception as e:
            print(B, env, "failed", e); continue
        b.set_inputs(h); b.run(); b.sync()
        t0 = time.perf_counter()
        for _ in range(3): b.run()
        b.sync()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print("B %5d %-40s strands %2d lanes %2d eval %8.3f ms  %.0f /s" % (B, env, b.strands, b.lanes, ms, B / ms * 1e3), flush=True)
        b.close()
