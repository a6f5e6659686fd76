"""Build schedules on the CPU box for a sweep of scheduler settings so that GPU time is spent on kernels only.
usage: python tools/prebuild_tapes.py <outdir> <row_overhead> ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
ov = sys.argv[2]
os.environ["CW_ROW_OVERHEAD"] = ov
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.sha256 import Sha256
from circom_amd.circuits.eddsa import SemaphoreStyle
d = os.path.join(out, "ov" + ov)
os.makedirs(d, exist_ok=True)
for name, prog, strands in (("poseidon2", Poseidon(2), (4,)), ("sha256_512", Sha256(512), (16,)), ("semaphore20", SemaphoreStyle(20), (4, 16))):
    if len(sys.argv) > 3 and name not in sys.argv[3:]:
        continue
    cp = compile_program(Program(prog), d, name, sym=False, strands=strands)
    print(ov, name, os.path.getsize(cp.tape_path))
