"""Diagnostics for the emitted 256-bit code: which signals of which circuit differ from the interpreter (run on a GPU box)."""
import os, sys, random, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from circom_amd import runtime as rt
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.opzoo import OperatorZoo, NAMES
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.basic import Multiplier2

def run(cp, rows, emitted, strands):
    os.environ["CW_FP_JIT"] = "1" if emitted else "0"
    os.environ["CW_STRANDS"] = str(strands)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    b = c.batch(len(rows))
    assert b.emitted == emitted, (b.emitted, emitted)
    b.set_inputs(rows)
    b.run(); b.sync()
    w, s = b.witnesses().copy(), b.status().copy()
    b.close(); c.close()
    return w, s

d = tempfile.mkdtemp()
which = sys.argv[1] if len(sys.argv) > 1 else "opzoo"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(1)
if which == "m2":
    cp = compile_program(Program(Multiplier2()), d, "m2", sym=False, fpjit=True)
    rows = [[rng.randrange(1 << 200), rng.randrange(1 << 200)] for _ in range(70)]
    names = None
elif which == "poseidon":
    cp = compile_program(Program(Poseidon(2)), d, "p2", sym=False, fpjit=True)
    q = cp.flat.fp.q
    rows = [[rng.randrange(q), rng.randrange(q)] for _ in range(70)]
    names = None
else:
    cp = compile_program(Program(OperatorZoo()), d, "opzoo", sym=False, fpjit=True)
    q = cp.flat.fp.q
    rows = [[rng.randrange(q), rng.randrange(1, 300)] for _ in range(70)] + [[rng.randrange(q), rng.randrange(q)] for _ in range(70)]
    names = NAMES
w1, s1 = run(cp, rows, True, S)
w0, s0 = run(cp, rows, False, S)
print("status emitted", [hex(x) for x in s1[::16]], "interp", s0[:8])
nw = w1.shape[1]
bad = {}
for i in range(len(rows)):
    for k in range(nw):
        if w1[i][k].tobytes() != w0[i][k].tobytes():
            bad.setdefault(k, []).append(i)
print("signals that differ:", len(bad), "of", nw)
for k in sorted(bad)[:40]:
    i = bad[k][0]
    nm = names[k - 1] if names and 1 <= k <= len(names) else ""
    print(k, nm, "n_bad", len(bad[k]), "inst", i, "got", hex(int.from_bytes(w1[i][k].tobytes(), "little")), "want", hex(int.from_bytes(w0[i][k].tobytes(), "little")))
