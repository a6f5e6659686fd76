"""Montgomery-form signals (lower.py pass A6, compiler.choose_mont): for arithmetic circuits the device value table holds
x R' mod q, so that a product of two run-time values is one Montgomery product instead of two; the runtime converts at its
boundary (ingest x R'^2, egress x 1, R1CS check mmul(A~, B~) == C~).  Results must be the canonical values of the reference.

CPU: the rewritten schedule, replayed with inputs x R' and outputs / R', gives the reference's signals for every variant;
mode choice; the multiplication count of Poseidon(2) drops from 1 071 to 828.
GPU: witnesses, single signals, .wtns bytes and the R1CS verdicts are identical with and without the rewrite."""
import os
import random
import struct

import numpy as np
import pytest

from circom_amd.compiler import compile_program, choose_mont
from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from circom_amd.circuits.basic import Num2Bits, Multiplier2
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.poseidon_constants import poseidon_hash
from circom_amd.circuits.sha256 import Sha256
from circom_amd.circuits.babyjub import ScalarMulBits, BASE8
from circom_amd.circuits.bigint import BigMod
from circom_amd.hip_elements import writers
from circom_amd.hip_elements.lower import lower
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, eval_tape
from test_pipe import _Mix, _inp

Q = PRIMES["bn128"]


def test_mode_choice():
    assert choose_mont(flatten(Program(Poseidon(2))))
    assert choose_mont(flatten(Program(ScalarMulBits(6))))
    assert choose_mont(flatten(Program(Multiplier2())))
    assert not choose_mont(flatten(Program(Num2Bits(32))))
    assert not choose_mont(flatten(Program(Sha256(8))))
    assert not choose_mont(flatten(Program(BigMod(16, 2))))               # run-time functions compute on canonical integers


def test_poseidon_multiplication_count():
    fc = flatten(Program(Poseidon(2)))
    plain, mont = lower(fc, n_strands=1), lower(fc, n_strands=1, mont=True)
    count = lambda t: 2 * t.stats["mul2"] + t.stats["mmul"] + t.stats["mulc"] + t.stats["linsum_terms"]
    assert (count(plain), count(mont)) == (1071, 828)
    assert mont.mont and not plain.mont and mont.stats["mont_conversions"] == 0


@pytest.mark.parametrize("kw", [dict(n_strands=1), dict(n_strands=4), dict(n_strands=16), dict(pipe=(8, 8)), dict(pipe=(4, 4))],
                         ids=["s1", "s4", "s16", "pipe88", "pipe44"])
def test_rewritten_schedules_replay_to_the_reference_signals(kw):
    rng = random.Random(11)
    cases = [(flatten(Program(Poseidon(2))), [[rng.randrange(Q), rng.randrange(Q)], [0, 0], [Q - 1, 1]]),
             (flatten(Program(_Mix())), [[rng.randrange(1 << 40), rng.randrange(Q), rng.randrange(Q)], [5, 9, 9], [0, 0, Q - 4],
                                         [(1 << 40) - 1, 7, 3]]),
             (flatten(Program(ScalarMulBits(5))), [[rng.randrange(2) for _ in range(5)] + list(BASE8)])]
    for fc, rows in cases:
        tp = lower(fc, mont=True, **kw)
        assert tp.mont
        for row in rows:
            want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, row))
            got, st = eval_tape(tp, _inp(fc, row))
            assert failed is None and st == 0 and got == want
    # a failing `===` is still reported (division by zero inside _Mix)
    fc = cases[1][0]
    got, st = eval_tape(lower(fc, mont=True, **kw), _inp(fc, [1, 2, Q - 3]))
    assert st & 0xFF == 1


def test_other_primes_and_refusals(tmp_path):
    for prime in ("bls12381", "pallas"):
        q = PRIMES[prime]
        fc = flatten(Program(Poseidon(2) if prime == "bn128" else _Mix(), prime=prime))
        row = [12345, q - 2, 17]
        want, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, row))
        got, st = eval_tape(lower(fc, n_strands=4, mont=True), _inp(fc, row))
        assert failed is None and st == 0 and got == want, prime
    with pytest.raises(ValueError):
        lower(flatten(Program(BigMod(16, 2))), mont=True)
    # variants of one tape must agree on the value form
    fc = flatten(Program(Poseidon(2)))
    with pytest.raises(AssertionError):
        writers.write_tape(str(tmp_path / "x.cwt"), [lower(fc, n_strands=1), lower(fc, n_strands=4, mont=True)])


def test_compile_program_marks_the_tape_and_the_loader_reads_the_flag(tmp_path):
    from circom_amd import runtime as rt
    for mont in (True, False):
        cp = compile_program(Program(Poseidon(2)), str(tmp_path), "p%d" % mont, sym=False, mont=mont)
        assert cp.tape.mont == mont
        c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
        assert c.montgomery == mont
        c.close()
    cp = compile_program(Program(Poseidon(2)), str(tmp_path), "auto", sym=False)
    assert cp.tape.mont
    cp = compile_program(Program(Num2Bits(20)), str(tmp_path), "bits", sym=False)
    assert not cp.tape.mont


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_same_witnesses_with_and_without_montgomery_form(tmp_path):
    from circom_amd import runtime as rt
    rng = random.Random(21)
    cases = [("poseidon2", Program(Poseidon(2)), lambda: [rng.randrange(Q), rng.randrange(Q)]),
             ("mix", Program(_Mix()), lambda: [rng.randrange(1 << 40), rng.randrange(Q), rng.randrange(Q)])]
    for name, prog, gen in cases:
        B = 200
        rows = [gen() for _ in range(B)]
        if name == "mix":
            rows[9] = [1, 2, Q - 3]                                   # a failing `===`
        res = {}
        for mont in (False, True):
            cp = compile_program(prog, str(tmp_path), "%s_%d" % (name, mont), sym=False, mont=mont, pipe=(8, 8))
            c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
            assert c.montgomery == mont
            for env in ({"CW_PIPE": "0", "CW_STRANDS": "1"}, {"CW_PIPE": "0", "CW_STRANDS": "16"}, {"CW_PIPE": "1"}):
                os.environ.update(env)
                try:
                    b = c.batch(B)
                finally:
                    for k in env:
                        del os.environ[k]
                b.set_inputs(rows)
                b.run(); b.check_r1cs(); b.sync()
                st = b.status()
                if name == "mix":
                    assert st[9] & rt.ST_ASSERT_FAILED
                    st = np.delete(st, 9)
                assert (st == 0).all(), (name, mont, env)
                w = b.witnesses()
                key = (mont, tuple(sorted(env.items())))
                res[key] = w.tobytes()
                fc = cp.flat
                for i in (0, 63, 64, 199):
                    if name == "mix" and i == 9:
                        continue
                    want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, rows[i]))
                    assert failed is None and w[i].tobytes() == b"".join(v.to_bytes(32, "little") for v in want), (name, mont, env, i)
                    assert b.signal(i, 1) == want[1] and b.witness(i) == want
                p = tmp_path / "w.wtns"
                b.write_wtns(3, p)
                assert p.read_bytes() == writers.wtns_bytes(Q, eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code,
                                                                         _inp(fc, rows[3]))[0])
                b.close()
            c.close()
        assert len(set(res.values())) == 1, name


@template
def _FlakyChain(c, n):
    # x[k+1] = x[k]^2 + b, but the witness code of link (a mod n) adds 1: every instance breaks a different constraint
    a = c.input("a")
    b = c.input("b")
    out = c.output("out")
    x = c.signal("x", n + 1)
    c.set(x[0], a + b)
    for k in range(n):
        c.hint(x[k + 1], x[k] * x[k] + b + (a % n).eq(k))
        c.enforce(x[k + 1], x[k] * x[k] + b, runtime_check=False)
    c.set(out, x[n] * 3 + x[1])


@pytest.mark.gpu
@pytest.mark.parametrize("mont", [False, True])
def test_gpu_r1cs_check_reports_the_same_first_violated_row_in_both_forms(tmp_path, mont):
    from circom_amd import runtime as rt
    n = 12
    cp = compile_program(Program(_FlakyChain(n)), str(tmp_path), "flaky", sym=False, mont=mont)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert c.montgomery == mont
    B = 150
    b = c.batch(B)
    b.set_inputs([[i, 1000 + i] for i in range(B)])
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    assert ((st & rt.ST_R1CS_FAILED) != 0).all() and ((st & rt.ST_ASSERT_FAILED) == 0).all()
    fb = b.r1cs_first_bad()
    names = {}
    for i in range(B):
        names.setdefault(i % n, set()).add(int(fb[i]))
    assert all(len(v) == 1 for v in names.values()) and len({next(iter(v)) for v in names.values()}) == n
    b.close(); c.close()
