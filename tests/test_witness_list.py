"""cw_set_witness_list: the egress of a batch hands out the witness of a SIMPLIFIED system (the reference's default --O1 keeps
a subset of the signals; its calculator writes those: witness2signal of the .dat, calcwit.hpp:54-56) while evaluation and
R1CS check stay on the full system.  CPU: the C ABI validates the list; GPU: every egress path against the oracle's full
witness read through the list."""
import os

import numpy as np
import pytest

from circom_amd import runtime as rt
from circom_amd.compiler import compile_program
from circom_amd.frontend.circom_exec import program_from_file
from circom_amd.frontend.circom_simplify import reduce_wtns, simplify_o1
from circom_amd.hip_elements.writers import wtns_bytes
from tests.test_circom_parity import LIB, SRC, _oracle, _rows


def _compiled(tmp_path, name, **kw):
    prog = program_from_file(os.path.join(SRC, name + ".circom"), [LIB])
    cp = compile_program(prog, str(tmp_path), "wl_" + name, sym=False, **kw)
    return cp, simplify_o1(cp.flat)


def test_the_c_abi_validates_the_list(tmp_path):
    cp, sm = _compiled(tmp_path, "sortpair", strands=(1,), fpjit=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert c.n_witness == 144 and c.n_public == 4
    for bad, msg in (([1, 2, 3, 4, 5], "must stay where they are"), ([0, 1, 2, 3, 4, 9, 7], "must increase"),
                     ([0, 1, 2, 3, 4, 144], "must increase and stay below"), ([0, 1, 2], "out of range"), ([0, 2, 3, 4, 5, 6], "must stay")):
        with pytest.raises(rt.CwError, match=msg):
            c.set_witness_list(bad)
    c.set_witness_list(sm.witness2signal)
    assert c.n_witness == sm.n_wires == 78 and c.n_constraints == 145           # the check still sees every row of the full system
    b = c.batch(3, device=-1)                                                     # (host-only batch: sizes follow the list)
    with pytest.raises(rt.CwError, match="live batches"):                         # a batch sized its images of the list at creation
        c.set_witness_list(list(range(c.n_signals)))
    b.close()
    c.set_witness_list(sm.witness2signal)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("sortpair", {}), ("sha256_64", {"bits": True})])
def test_gpu_egress_follows_the_list(tmp_path, name, kw):
    cp, sm = _compiled(tmp_path, name, **kw)
    fc = cp.flat
    w2s = sm.witness2signal
    n = 70
    if name == "sha256_64":
        rng = np.random.default_rng(4)
        rows = [[int(x) for x in rng.integers(0, 2, 64)] for _ in range(n)]
    else:
        rows = _rows(name, fc, n, 2)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    c.set_witness_list(w2s)
    b = c.batch(n)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i in (0, 33, n - 1):
        full = _oracle(fc, rows[i])
        assert b.witness(i) == [full[s] for s in w2s]
        b.write_wtns(i, tmp_path / "w.wtns")
        assert (tmp_path / "w.wtns").read_bytes() == reduce_wtns(wtns_bytes(fc.fp.q, full), w2s)
    bulk = b.witnesses(0, n)
    assert bulk.shape == (n, len(w2s), 32)
    full = _oracle(fc, rows[7])
    assert [int.from_bytes(bulk[7, k].tobytes(), "little") for k in range(len(w2s))] == [full[s] for s in w2s]
    # a witness that violates a row the SIMPLIFIED system no longer has is still caught: the check runs on the full system
    b.close(); c.close()
