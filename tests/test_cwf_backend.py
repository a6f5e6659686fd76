"""The front-end / back-end seam as files: a flat circuit written to `.cwf` and lowered by the back-end
process gives byte-identical artefacts to the in-process path, for an arithmetic circuit, a bit-level one, one with
run-time functions and one with a Mixed component cluster."""
import filecmp
import os
import subprocess
import sys
from pathlib import Path

import pytest

from circom_amd.compiler import compile_program
from circom_amd.cwf import write_cwf, read_cwf
from circom_amd.frontend.dsl import Program

ROOT = Path(__file__).resolve().parent.parent


def _cases():
    from circom_amd.circuits.poseidon import Poseidon
    from circom_amd.circuits.sha256 import Sha256
    from circom_amd.circuits.basic import MixedArray
    from circom_amd.circuits.bigint import BigMultModP
    yield "poseidon2", lambda: Program(Poseidon(2)), {}
    yield "sha256_64", lambda: Program(Sha256(64)), {"bits": True}
    yield "mixed", lambda: Program(MixedArray(((2, 3), (1, 5)))), {}
    yield "bigmult", lambda: Program(BigMultModP(16, 2), prime="bls12381"), {}


@pytest.mark.parametrize("name,mk,kw", list(_cases()), ids=[c[0] for c in _cases()])
def test_cwf_round_trip_and_backend_process(name, mk, kw, tmp_path):
    cp = compile_program(mk(), str(tmp_path / "a"), name, sym=False, strands=(1,), **kw)
    fc = cp.flat
    cwf = tmp_path / (name + ".cwf")
    write_cwf(cwf, fc)
    back = read_cwf(cwf)
    assert back.n_signals == fc.n_signals and back.constants == list(fc.constants) and back.inputs == list(fc.inputs)
    assert all((back.code[c] == fc.code[c]).all() for c in back.code)
    assert back.constraints == [tuple(dict(p) for p in cons) for cons in fc.constraints]
    assert back.io_map == list(fc.io_map)
    env = dict(os.environ)
    if kw.get("bits"):
        env["CW_JIT"] = "0"        # (the emitted bit-plane code of this circuit is a minute of assembling that this test does not look at)
    r = subprocess.run([sys.executable, "-m", "circom_amd.hip_backend", str(cwf), "-o", str(tmp_path / "b"), "--strands", "1"],
                       capture_output=True, text=True, cwd=str(ROOT), timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    for ext in (".dat", ".r1cs"):
        assert filecmp.cmp(tmp_path / "a" / (name + ext), tmp_path / "b" / (name + ext), shallow=False), ext
    if kw.get("bits") is None:                 # (bits=True is a caller's choice; the process applies the auto rule)
        assert filecmp.cmp(tmp_path / "a" / (name + ".cwt"), tmp_path / "b" / (name + ".cwt"), shallow=False)
    from circom_amd import runtime as rt
    rt.Circuit(tmp_path / "b" / (name + ".cwt"), tmp_path / "b" / (name + ".dat"), tmp_path / "b" / (name + ".r1cs")).close()
