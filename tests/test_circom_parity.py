"""Circuits compiled from circom SOURCE TEXT against the reference's own runtime and on the GPU.

CPU: oracle/emit_ref_cpp.py prints the flat circuit the text front-end produced in the reference's emission format
(`<name>.cpp` + `.dat`), the REFERENCE runtime (common/main.cpp + calcwit.cpp + generic/fr.cpp, built by oracle/Makefile
into oracle/_ref) executes it, and its `.wtns` files equal the oracle's byte for byte - including a function that the
text front-end compiled to tier-2 bytecode (the reference runs it as C++ control flow over its own Fr_* calls).
The GPU side of the same circuits: tests/test_zz_circom_gpu.py."""
import hashlib
import os
import random

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.circom_exec import program_from_file
from circom_amd.hip_elements.writers import wtns_bytes
from oracle.tape_eval import eval_flat

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(ROOT, "circom_amd", "circuits", "circomlib")
SRC = os.path.join(HERE, "circom")


@pytest.fixture(scope="module")
def libs(tmp_path_factory):
    from circom_amd.circuits.poseidon_constants import circom_text
    from oracle.field import PRIMES
    out = [LIB]
    for prime in ("bn128", "bls12381"):
        d = tmp_path_factory.mktemp("poseidon_" + prime)
        (d / "poseidon_constants.circom").write_text(circom_text(PRIMES[prime]))
        out.append(str(d))
    return out


def _libs_for(libs, prime):
    return [libs[0], libs[1] if prime == "bn128" else libs[2]]


def _oracle(fc, vals):
    inp = {fc.main_input_start + k: int(v) for k, v in enumerate(vals)}
    sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp, functions=fc.functions)
    assert failed is None
    return sig


def _rows(name, fc, n, seed):
    rng = random.Random(seed)
    q = fc.fp.q
    if name == "bigmultmodp":
        rows = []
        for t in range(n):
            p = [rng.getrandbits(32) for _ in range(3)]
            p[2] |= (1 << 31) if t % 2 else 1
            rows.append([rng.getrandbits(32) for _ in range(6)] + p)
        return rows
    if name == "modinv":
        p = 2147483647                       # 2^31 - 1 on two 16-bit limbs
        lim = lambda x: [x & 0xFFFF, x >> 16]
        return [lim(a) + lim(p) for a in [0, 1, p - 1] + [rng.randrange(1, p) for _ in range(max(0, n - 3))]]
    if name == "bigmult_style":
        rows = []
        lim = lambda x: [(x >> (28 * i)) & ((1 << 28) - 1) for i in range(3)]
        for t in range(n):
            p = (rng.getrandbits(84) | (1 << 83)) if t % 2 else (rng.getrandbits(64) | (1 << 56))
            rows.append(lim(rng.randrange(p)) + lim(rng.randrange(p)) + lim(p))
        return rows
    if name == "multiand5":
        return [[1] * 5, [1, 1, 0, 1, 1]] + [[rng.randrange(2) for _ in range(5)] for _ in range(max(0, n - 2))]
    if name == "sortpair":
        return [[rng.getrandbits(16), rng.getrandbits(16)] for _ in range(n - 1)] + [[777, 777]]
    return [[rng.randrange(q) for _ in range(fc.n_main_inputs)] for _ in range(n)]


@pytest.mark.parametrize("name,prime", [("sortpair", "bn128"), ("poseidon2", "bls12381"), ("bigmultmodp", "bls12381"),
                                        ("opzoo", "bn128"), ("modinv", "bls12381"), ("multiand5", "bn128"),
                                        ("bigmult_style", "bls12381")])
def test_reference_runtime_executes_circuits_compiled_from_text(name, prime, libs, tmp_path):
    from oracle import ref_build
    if not os.path.isdir(os.path.join(os.path.dirname(ref_build.__file__), "_ref", prime)) and not ref_build.REF_ROOT.exists():
        pytest.skip("no reference build for " + prime)
    prog = program_from_file(os.path.join(SRC, name + ".circom"), _libs_for(libs, prime), prime=prime)
    cp = compile_program(prog, str(tmp_path), "txt_%s_%s" % (name, prime), sym=False, strands=(1,), fpjit=False)
    fc = cp.flat
    rows = _rows(name, fc, 6, 11)
    if name == "opzoo":
        q = fc.fp.q
        rows += [[3, 11], [0, 0], [q - 1, q - 1], [(q >> 1) + 1, 255], [1 << 200, q - 3]]
    ref_build.build_circuit(cp)
    raw = b"".join(int(v).to_bytes(32, "little") for r in rows for v in r)
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=str(tmp_path / "w_"))
    for i, r in enumerate(rows):
        assert (tmp_path / ("w_%d.wtns" % i)).read_bytes() == wtns_bytes(fc.fp.q, _oracle(fc, r)), (name, i)
