python - <<'PY'
import sys, tempfile, time
sys.path.insert(0,'.')
import numpy as np, torch
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.sha256 import Sha256
from circom_amd import runtime as rt
from bench import synth_inputs
import os
d=tempfile.mkdtemp()
for name, prog, B, strands in (("poseidon2", Program(Poseidon(2)), 65536, (2,3,4,6)), ("sha256_512", Program(Sha256(512)), 4096, (8,12,16)), ("sha256_512", None, 8192, (8,16))):
    if prog is not None:
        cp=compile_program(prog, d, name, sym=False, strands=strands)
    c=rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    h=synth_inputs(name, c.q, B, c.n_inputs, 1)
    din=torch.from_numpy(h).cuda()
    for S in strands:
        os.environ["CW_STRANDS"]=str(S)
        b=c.batch(B)
        b.set_inputs_device(din.data_ptr())
        for _ in range(2): b.run(); b.check_r1cs()
        b.sync()
        e=[torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); 
        for _ in range(3): b.run()
        e[1].record()
        for _ in range(3): b.check_r1cs()
        e[2].record(); torch.cuda.synchronize()
        print("%s B=%d S=%d (picked %d): eval %.3f ms r1cs %.3f ms bad=%d"%(name,B,S,b.strands,e[0].elapsed_time(e[1])/3, e[1].elapsed_time(e[2])/3, int((b.status()!=0).sum())))
        b.close()
    c.close()
PY
