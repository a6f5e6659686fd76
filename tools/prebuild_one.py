"""python tools/prebuild_one.py <outdir> <name> <strands,comma>  (name: poseidon2 | sha256_512 | semaphore20)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
out, name, strands = sys.argv[1], sys.argv[2], tuple(int(x) for x in sys.argv[3].split(","))
if name == "poseidon2":
    from circom_amd.circuits.poseidon import Poseidon as T; prog = T(2)
elif name == "sha256_512":
    from circom_amd.circuits.sha256 import Sha256 as T; prog = T(512)
else:
    from circom_amd.circuits.eddsa import SemaphoreStyle as T; prog = T(20, name.endswith("p"))
os.makedirs(out, exist_ok=True)
pipe = tuple(int(x) for x in os.environ["CW_PIPE_SHAPE"].split(",")) if os.environ.get("CW_PIPE_SHAPE") else None
cp = compile_program(Program(prog), out, name, sym=False, strands=strands, pipe=pipe)
print(out, name, strands, cp.tape.stats.get("barriers"))
