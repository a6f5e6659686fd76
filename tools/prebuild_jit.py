"""Lower a SHA-256 workload WITH the emitted-code section into gpurun_in/jit (travels with gpurun):
   python tools/prebuild_jit.py <message bits> [prefetch]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.sha256 import Sha256

nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_in", "jit")
os.makedirs(out, exist_ok=True)
t0 = time.time()
cp = compile_program(Program(Sha256(nbits)), out, "sha256_%d" % nbits, sym=False, strands=(1,), bits=True, jit=True)
print("compiled in %.0f s" % (time.time() - t0), json.dumps(cp.jit.stats), "code bytes", len(cp.jit.code))
json.dump(cp.jit.stats, open(os.path.join(out, "sha256_%d.jitstats.json" % nbits), "w"))
