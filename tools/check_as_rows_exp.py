"""Experiment: what would the R1CS check cost if it ran as ROWS of the schedule engine (Montgomery-form dot products with one
reduction, strands, LDS hand-over) instead of the term-stream kernel?  The check program - per constraint: linear combinations,
one product, one equality assertion - is appended to the witness code and lowered with it; the growth of the evaluation
kernel's time is the price of the check in that engine.   python tools/check_as_rows_exp.py [workload] [batch]"""
import copy
import os
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import bench
from circom_amd import opcodes as O
from circom_amd import runtime as rt
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements import writers
from circom_amd.hip_elements.lower import lower
from circom_amd import compiler


def check_program(fc):
    """flat ops (op, dk, dv, ak, av, bk, bv) computing A*B - C per constraint into temps + ASSERT_EQ rows"""
    q = fc.fp.q
    consts = list(fc.constants)
    cid = {v: i for i, v in enumerate(consts)}

    def const(v):
        v %= q
        if v not in cid:
            cid[v] = len(consts)
            consts.append(v)
        return cid[v]

    ops = []
    nt = [fc.n_temps]

    def tmp():
        nt[0] += 1
        return nt[0] - 1

    def lin(part):
        acc = None
        for s, cf in sorted(part.items()):
            cf %= q
            if s == 0:
                term = (O.K_CONST, const(cf))
            elif cf == 1:
                term = (O.K_SIG, s)
            else:
                t = tmp()
                ops.append((O.MUL, O.K_TMP, t, O.K_CONST, const(cf), O.K_SIG, s))
                term = (O.K_TMP, t)
            if acc is None:
                acc = term
            else:
                t = tmp()
                ops.append((O.ADD, O.K_TMP, t, acc[0], acc[1], term[0], term[1]))
                acc = (O.K_TMP, t)
        return acc or (O.K_CONST, const(0))

    for a, b, c in fc.constraints:
        if not a or not b:
            if len(c) == 2 and 0 not in c and sorted(v % q for v in c.values()) in ([1, q - 1],):
                (x, _), (y, _) = sorted(c.items())
                ops.append((O.ASSERT_EQ, O.K_NONE, 0, O.K_SIG, x, O.K_SIG, y))
            else:
                lc = lin(c)
                ops.append((O.ASSERT_EQ, O.K_NONE, 0, lc[0], lc[1], O.K_CONST, const(0)))
            continue
        la, lb, lc = lin(a), lin(b), lin(c)
        t = tmp()
        ops.append((O.MUL, O.K_TMP, t, la[0], la[1], lb[0], lb[1]))
        ops.append((O.ASSERT_EQ, O.K_NONE, 0, O.K_TMP, t, lc[0], lc[1]))
    return ops, consts, nt[0]


name = sys.argv[1] if len(sys.argv) > 1 else "poseidon2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
fc = flatten(bench.make_program(name))
ops, consts, n_temps = check_program(fc)
fc2 = copy.copy(fc)
code = {k: v.copy() for k, v in fc.code.items()}
arr = np.asarray(ops, dtype=np.int64)
add = {"op": arr[:, 0], "dk": arr[:, 1], "dv": arr[:, 2], "ak": arr[:, 3], "av": arr[:, 4], "bk": arr[:, 5], "bv": arr[:, 6],
       "ck": np.full(len(ops), O.K_NONE), "cv": np.zeros(len(ops), dtype=np.int64)}
fc2.code = {k: np.concatenate([code[k].astype(np.int64), add[k].astype(np.int64)]) for k in code}
fc2.constants = consts
fc2.n_temps = n_temps
strands = compiler.strands_for(B)
mont = compiler.choose_mont(fc)
d = tempfile.mkdtemp()
res = {}
for tag, f in (("witness only", fc), ("witness + check rows", fc2)):
    t = [lower(f, n_strands=s, mont=mont) for s in strands]
    p = os.path.join(d, tag.replace(" ", "_").replace("+", "p"))
    writers.write_tape(p + ".cwt", t, None)
    writers.write_dat(p + ".dat", f)
    writers.write_r1cs(p + ".r1cs", fc)
    c = rt.Circuit(p + ".cwt", p + ".dat", p + ".r1cs")
    h = bench.synth_inputs(name, c.q, B, c.n_inputs, 3)
    b = c.batch(B)
    b.set_inputs(h)
    b.run(); b.sync()
    assert (b.status() == 0).all()
    t0 = time.perf_counter()
    for _ in range(5):
        b.run()
    b.sync()
    ev = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    for _ in range(5):
        b.check_r1cs()
    b.sync()
    ck = (time.perf_counter() - t0) / 5 * 1e3
    print("%-22s rows %6d mmul %6d strands %d: evaluation %.3f ms   term-stream check %.3f ms" % (tag, t[0].stats["rows"], c.n_mmul, b.strands, ev, ck), flush=True)
    res[tag] = ev
    b.close(); c.close()
print("check as rows: +%.3f ms" % (res["witness + check rows"] - res["witness only"]))
