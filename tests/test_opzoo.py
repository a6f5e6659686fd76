"""Every operator of the witness-code expression language, end to end: the Python oracle, the lowered schedule
and the HIP path against the reference's own runtime (its Fr_* functions, generic/fr.cpp) on edge and random
operands — the circuit-level counterpart of the per-operator device tests."""
import random

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.opzoo import OperatorZoo, NAMES
from circom_amd.hip_elements.lower import lower
from circom_amd.hip_elements.writers import wtns_bytes
from oracle import ref_build
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, eval_tape


def _operands(q, n_random, seed):
    half = q >> 1
    edges = [0, 1, 2, 3, 31, 32, 63, 64, 65, 253, 254, 255, 256, 257, half, half + 1, half - 1, q - 1, q - 2, q - 253,
             q - 254, q - 255, q - 64, q - 1 - (1 << 200), (1 << 31) - 1, 1 << 31, 1 << 32, (1 << 64) - 1, 1 << 64,
             1 << 128, 1 << 253]
    edges = sorted({x % q for x in edges})         # canonical operands only (2^253 exceeds the 253-bit bls12377 prime)
    rng = random.Random(seed)
    return ([[x, y] for x in edges for y in edges] + [[rng.randrange(q), rng.randrange(q)] for _ in range(n_random)] +
            [[rng.randrange(q), rng.randrange(300)] for _ in range(n_random // 3)])


BIG_PRIMES = ["bn128", "bls12381", "bls12377", "grumpkin", "pallas", "vesta", "secq256r1"]   # constants.rs:3-13


@pytest.mark.parametrize("prime", BIG_PRIMES)
def test_oracle_and_schedule_match_reference_runtime_on_every_operator(tmp_path, prime):
    from conftest import ensure_ref
    ensure_ref(prime)                       # renders generic/fr.cpp for this prime (secq256r1: cannotOptimize variant)
    q = PRIMES[prime]
    cp = compile_program(Program(OperatorZoo(), prime=prime), str(tmp_path), "opzoo", sym=False)
    try:
        ref_build.build_circuit(cp)
    except RuntimeError as e:
        pytest.skip(str(e))
    fc = cp.flat
    rows = _operands(q, 150, 5)
    raw = b"".join(v.to_bytes(32, "little") for r in rows for v in r)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=pre)
    tapes = [lower(fc, n_strands=S) for S in (1, 4)]
    for i, r in enumerate(rows):
        inp = {fc.main_input_start: r[0], fc.main_input_start + 1: r[1]}
        sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None
        ref = open(pre + "%d.wtns" % i, "rb").read()
        if ref != wtns_bytes(q, sig):
            refv = [int.from_bytes(ref[76 + 32 * k:108 + 32 * k], "little") for k in range(len(sig))]
            raise AssertionError((hex(r[0]), hex(r[1]), [NAMES[k - 1] for k in range(1, 1 + len(NAMES)) if refv[k] != sig[k]]))
        if i % 29 == 0:
            for t in tapes:
                got, st = eval_tape(t, inp)
                assert st == 0 and got == sig


@pytest.mark.gpu
@pytest.mark.parametrize("prime", BIG_PRIMES)
def test_gpu_matches_oracle_on_every_operator(tmp_path, prime):
    from circom_amd import runtime as rt
    q = PRIMES[prime]
    cp = compile_program(Program(OperatorZoo(), prime=prime), str(tmp_path), "opzoo", sym=False)
    fc = cp.flat
    rows = _operands(q, 600, 6)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    b = c.batch(len(rows))
    b.set_inputs(rows)
    b.run(); b.sync()
    assert (b.status() == 0).all()
    got = b.witnesses()
    for i, r in enumerate(rows):
        inp = {fc.main_input_start: r[0], fc.main_input_start + 1: r[1]}
        sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        if got[i].tobytes() != b"".join(v.to_bytes(32, "little") for v in sig):
            w = [int.from_bytes(got[i][k].tobytes(), "little") for k in range(len(sig))]
            raise AssertionError((hex(r[0]), hex(r[1]), [NAMES[k - 1] for k in range(1, 1 + len(NAMES)) if w[k] != sig[k]]))
    b.close(); c.close()
