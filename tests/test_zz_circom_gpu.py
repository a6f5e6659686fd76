"""Circuits compiled from circom SOURCE TEXT on the GPU, through the C ABI: witness bytes == oracle, R1CS check green
(the CPU side - reference runtime executing the same circuits - is tests/test_circom_parity.py)."""
import hashlib
import os

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.circom_exec import program_from_file
from circom_amd.hip_elements.writers import wtns_bytes
from tests.test_circom_parity import SRC, _libs_for, _oracle, _rows, libs  # noqa: F401  (libs is a fixture)


def _gpu_batch(cp, rows):
    from circom_amd import runtime as rt
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    b = c.batch(len(rows))
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    return c, b


@pytest.mark.gpu
@pytest.mark.parametrize("name,prime,n", [("sortpair", "bn128", 200), ("poseidon2", "bn128", 300), ("bigmultmodp", "bls12381", 96)])
def test_gpu_runs_circuits_compiled_from_text(name, prime, n, libs, tmp_path):
    prog = program_from_file(os.path.join(SRC, name + ".circom"), _libs_for(libs, prime), prime=prime)
    cp = compile_program(prog, str(tmp_path), "txt_" + name, sym=False)
    fc = cp.flat
    rows = _rows(name, fc, n, 5)
    c, b = _gpu_batch(cp, rows)
    assert (b.status() == 0).all()
    for i in (0, 1, n // 2, n - 1):
        assert b.witness(i) == _oracle(fc, rows[i]), (name, i)
        p = tmp_path / ("g%d.wtns" % i)
        b.write_wtns(i, p)
        assert p.read_bytes() == wtns_bytes(fc.fp.q, _oracle(fc, rows[i]))
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_sha256_from_text_through_the_bit_plane_engine(libs, tmp_path):
    """Sha256(64) written in circom (circuits/circomlib/sha256/*.circom): bit-plane program, digests against hashlib, one
    golden-style full witness against the oracle"""
    prog = program_from_file(os.path.join(SRC, "sha256_64.circom"), libs[:2])
    cp = compile_program(prog, str(tmp_path), "txt_sha256_64", sym=False, bits=True)
    fc = cp.flat
    rng = np.random.default_rng(9)
    n = 96
    msgs = [rng.bytes(8) for _ in range(n)]
    rows = [[(m[k // 8] >> (7 - k % 8)) & 1 for k in range(64)] for m in msgs]
    c, b = _gpu_batch(cp, rows)
    assert b.bitmode and (b.status() == 0).all()
    for i in (0, 31, 32, 95):
        digest = hashlib.sha256(msgs[i]).digest()
        assert [b.signal(i, 1 + k) for k in range(256)] == [(digest[k // 8] >> (7 - k % 8)) & 1 for k in range(256)]
    assert b.witness(33) == _oracle(fc, rows[33])
    b.close(); c.close()
