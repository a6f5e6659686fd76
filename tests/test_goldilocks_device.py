"""The 64-bit runtime ON THE DEVICE (`--prime goldilocks`, csrc/cw64.hip + hip_elements/lower64.py): SURVEY row f4.

The golden `.wtns` files come from the reference's own 64-bit runtime (`common64/` + `goldilocks/fr.hpp`, built by
oracle/Makefile circuit64; tests/golden/reference_wtns_goldilocks.json, whose generator also pins the oracle:
tests/test_goldilocks_oracle.py): five circuits incl. the operator zoo (every operator of the witness language on edge and
random operands) and Poseidon(2) over Goldilocks.  CPU: the tape / .dat / .r1cs of the 64-bit formats load through the C ABI
and damaged ones are refused.  GPU: every golden vector byte for byte (n8 = 8 files), the R1CS check clean on them and firing on
a corrupted witness, random batches against the oracle."""
import hashlib
import json
import os
import random
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_golden import goldilocks_cases                                            # noqa: E402

from circom_amd.compiler import compile_program                                      # noqa: E402
from circom_amd.hip_elements.writers import wtns_bytes                              # noqa: E402
from oracle.tape_eval import eval_flat, check_r1cs                                   # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_wtns_goldilocks.json")))["cases"]
CASES = goldilocks_cases()
Q = 18446744069414584321


def test_64_bit_artefacts_load_through_the_c_abi(tmp_path):
    from circom_amd import runtime as rt
    mk, rows = CASES["poseidon2"]
    cp = compile_program(mk(), str(tmp_path), "poseidon2", sym=False)
    assert open(cp.tape_path, "rb").read(4) == b"CW64"
    # .dat of the 64-bit runtime: hash map + witness list + io map, NO constants section (c_code_generator.rs:838-841)
    from circom_amd.hip_elements.writers import dat_io_map
    assert os.path.getsize(cp.dat_path) == 256 * 24 + cp.flat.n_signals * 8 + len(dat_io_map(cp.flat.io_map))
    r1 = open(cp.r1cs_path, "rb").read()
    assert struct.unpack_from("<I", r1, r1.index(struct.pack("<IQ", 1, 4 + 8 + 16 + 8 + 4)) + 12)[0] == 8      # field size 8 in the header section
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert (c.q, c.n_signals, c.n_constraints) == (Q, cp.flat.n_signals, len(cp.flat.constraints))
    b = c.batch(3, device=-1)                                     # host-only: staging and its error semantics work, nothing computes
    b.set_inputs_json(0, '{"inputs": ["1", "2"]}')
    assert b.remaining_inputs(0) == 0 and b.staged_input(0, 1) == 2
    with pytest.raises(rt.CwError):
        b.run()
    b.close(); c.close()
    raw = bytearray(open(cp.tape_path, "rb").read())
    for mutate in (lambda t: t.__setitem__(slice(8, 16), struct.pack("<Q", Q - 2)),          # another prime
                   lambda t: t.__setitem__(slice(len(t) - 32, len(t) - 28), struct.pack("<I", 200)),     # unknown opcode in the last row
                   lambda t: t.__delitem__(slice(len(t) - 8, len(t)))):                     # truncated
        t2 = bytearray(raw)
        mutate(t2)
        (tmp_path / "bad.cwt").write_bytes(bytes(t2))
        with pytest.raises(rt.CwError):
            rt.Circuit(tmp_path / "bad.cwt", cp.dat_path, cp.r1cs_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_gpu_reproduces_the_64_bit_runtimes_wtns(name, tmp_path):
    from circom_amd import runtime as rt
    mk, rows = CASES[name]
    vecs = GOLD[name]["vectors"]
    cp = compile_program(mk(), str(tmp_path), name, sym=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    b = c.batch(len(vecs))
    b.set_inputs([[int(v) for v in vec["inputs"]] for vec in vecs])
    b.run()
    if c.n_constraints:
        b.check_r1cs()
    b.sync()
    assert (b.status() == 0).all(), b.status()
    for i, vec in enumerate(vecs):
        p = tmp_path / ("g%d.wtns" % i)
        b.write_wtns(i, p)
        got = p.read_bytes()
        assert len(got) == vec["wtns_len"] and hashlib.sha256(got).hexdigest() == vec["wtns_sha256"], (name, i)
        if vec["wtns_hex"]:
            assert got.hex() == vec["wtns_hex"]
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_64_bit_random_batch_and_r1cs_check(tmp_path):
    """Poseidon(2) over Goldilocks x 300 and the operator zoo x 300 against the oracle; a witness corrupted through a wrong
    hint must be caught by the check with the oracle's first bad row"""
    from circom_amd import runtime as rt
    from circom_amd.frontend.dsl import Program, template
    rnd = random.Random(9)
    for name in ("poseidon2", "opzoo"):
        mk, _ = CASES[name]
        cp = compile_program(mk(), str(tmp_path), name, sym=False)
        fc = cp.flat
        c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
        B = 300
        rows = [[rnd.choice((0, 1, Q - 1, Q >> 1, (Q >> 1) + 1, rnd.randrange(Q), rnd.randrange(1 << 20), 63, 64, Q - 64)) for _ in range(c.n_inputs)] for _ in range(B)]
        b = c.batch(B)
        b.set_inputs(rows)
        b.run()
        if c.n_constraints:
            b.check_r1cs()
        b.sync()
        st = b.status()
        pub = b.public_signals()
        for i in range(B):
            sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start + k: v for k, v in enumerate(rows[i])})
            assert (failed is None) == ((st[i] & 3) == 0), (name, i, rows[i])
            if failed is None:
                assert b.witness(i) == sig, (name, i, rows[i])
                assert [int.from_bytes(pub[i, k].tobytes(), "little") for k in range(c.n_public)] == sig[1:1 + c.n_public]
            else:
                assert st[i] >> 8 == failed
        assert b.signal(7, 1) == b.witness(7)[1]
        b.close(); c.close()

    @template
    def BadSquare(cx):
        a = cx.input("a")
        out = cx.output("out")
        cx.hint(out, a * a + (a & 1))                             # wrong for odd a
        cx.enforce(out, a * a, runtime_check=False)
    cp = compile_program(Program(BadSquare(), prime="goldilocks"), str(tmp_path), "badsq", sym=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    rows = [[rnd.randrange(Q)] for _ in range(100)]
    b = c.batch(100)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st, fb = b.status(), b.r1cs_first_bad()
    for i, (a,) in enumerate(rows):
        assert bool(st[i] & 4) == bool(a & 1) and (fb[i] == 0) == bool(a & 1), i
    b.close(); c.close()
