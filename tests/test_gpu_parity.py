"""GPU parity tests (run on a real MI355X through the C ABI): HIP path vs the CPU oracle, bit-exact."""
import os
import random

import numpy as np
import pytest

from circom_amd import runtime as rt
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program, template
from circom_amd.circuits.basic import Multiplier2, BasicMain, IsZero, Num2Bits
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.poseidon_constants import poseidon_hash
from circom_amd.hip_elements import lower as L
from circom_amd.hip_elements.writers import wtns_bytes
from oracle.field import Field, PRIMES, FieldError
from oracle.tape_eval import eval_flat, check_r1cs

pytestmark = pytest.mark.gpu


def _edges(f):
    q = f.q
    e = [0, 1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 253, 254, 255, 256, f.half, f.half + 1, f.half - 1, q - 1, q - 2,
         (1 << 31) - 1, 1 << 31, (1 << 32) - 1, 1 << 32, 1 << 63, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 128,
         q - (1 << 31), q - 253, q - 254, q - 255, q - 64, q - 32, q - 1 - (1 << 200), 1 << (f.bits - 1),
         (1 << (f.bits - 1)) - 1, f.mask % q, (1 << 224) - 1, 1 << 224]
    return [x % q for x in e]


@pytest.mark.parametrize("prime", ["bn128", "bls12381", "bls12377", "grumpkin", "pallas", "vesta", "secq256r1"])
def test_device_field_ops_match_oracle(prime):
    f = Field(PRIMES[prime])
    rng = random.Random(7)
    edges = _edges(f)
    rand = [rng.randrange(f.q) for _ in range(300)] + [rng.randrange(1 << 40) for _ in range(50)]
    A = [a for a in edges for _ in edges] + [rng.choice(rand + edges) for _ in range(3000)]
    B = [b for _ in edges for b in edges] + [rng.choice(rand + edges) for _ in range(3000)]
    Cc = [rng.choice(rand + edges) for _ in A]
    rinv = pow(1 << 261, -1, f.q)          # device Montgomery radix R' = 2^261 (csrc/fp256.hip.h)
    ops = {L.D_ADD: f.add, L.D_SUB: f.sub, L.D_MMUL: lambda a, b: a * b * rinv % f.q, L.D_MUL2: f.mul, L.D_SHL: f.shl, L.D_SHR: f.shr, L.D_BAND: f.band,
           L.D_BOR: f.bor, L.D_BXOR: f.bxor, L.D_LT: f.lt, L.D_GT: f.gt, L.D_LEQ: f.leq, L.D_GEQ: f.geq, L.D_EQ: f.eq,
           L.D_NEQ: f.neq, L.D_LAND: f.land, L.D_LOR: f.lor, L.D_POW: f.pow}
    for dop, fn in ops.items():
        got, st = rt.fp_op(f.q, dop, A, B, Cc)
        want = [fn(a, b) for a, b in zip(A, B)]
        bad = [i for i in range(len(A)) if got[i] != want[i]]
        assert not bad, (prime, L.D_NAMES[dop], hex(A[bad[0]]), hex(B[bad[0]]), hex(got[bad[0]]), hex(want[bad[0]]))
    got, st = rt.fp_op(f.q, L.D_MUL2, A, A, Cc)          # every lane multiplies a value by itself: the squaring path
    assert got == [a * a % f.q for a in A], (prime, "mul2 square")
    for dop, fn in {L.D_NEG: f.neg, L.D_BNOT: f.bnot, L.D_LNOT: f.lnot, L.D_INV: f.inv, L.D_COPY: lambda x: x}.items():
        got, st = rt.fp_op(f.q, dop, A, B, Cc)
        want = [fn(a) for a in A]
        assert got == want, (prime, L.D_NAMES[dop])
    for dop, fn in {L.D_IDIV: f.idiv, L.D_MOD: f.mod}.items():
        got, st = rt.fp_op(f.q, dop, A, B, Cc)
        for i, (a, b) in enumerate(zip(A, B)):
            if b == 0:
                assert st[i] == rt.ST_ARITH
            else:
                assert st[i] == 0 and got[i] == fn(a, b), (prime, L.D_NAMES[dop], hex(a), hex(b))
    got, st = rt.fp_op(f.q, L.D_MADD, A, B, Cc)
    assert got == [(a * b * rinv + c) % f.q for a, b, c in zip(A, B, Cc)]
    got, st = rt.fp_op(f.q, L.D_SELECT, A, B, Cc)
    assert got == [b if a else c for a, b, c in zip(A, B, Cc)]
    got, st = rt.fp_op(f.q, L.D_ASSERT_EQ, A, B, Cc)
    assert list(st) == [0 if a == b else rt.ST_ASSERT_FAILED for a, b in zip(A, B)]


@pytest.mark.parametrize("prime", ["bn128", "bls12381"])
def test_device_short_path_products(prime):
    """Waves whose 64 lanes all hold signed-small operands take the 64x64-bit short path (fe_mul2_auto):
    same canonical result as the Montgomery path, including the +-(2^64-1) boundaries, zero and mixed signs;
    one large lane anywhere in a wave sends that wave down the generic path."""
    f = Field(PRIMES[prime])
    q = f.q
    small = [0, 1, 2, 3, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, 1 << 32, (1 << 63) - 1, 1 << 63, (1 << 64) - 1,
             q - 1, q - 2, q - (1 << 32), q - (1 << 63), q - ((1 << 64) - 1)]
    edge = [1 << 64, q - (1 << 64), q >> 1, 12345678901234567890123456789]       # not signed-small
    rng = random.Random(3)
    A = [a for a in small for _ in small]                 # 256 lanes = 4 waves, all signed-small
    B = [b for _ in small for b in small]
    A += [rng.choice(small) for _ in range(192)] + [rng.choice(small + edge) for _ in range(320)]
    B += [rng.choice(small) for _ in range(192)] + [rng.choice(small + edge) for _ in range(320)]
    got, st = rt.fp_op(q, L.D_MUL2, A, B, A)
    want = [f.mul(a, b) for a, b in zip(A, B)]
    bad = [i for i in range(len(A)) if got[i] != want[i]]
    assert not bad, (prime, hex(A[bad[0]]), hex(B[bad[0]]), hex(got[bad[0]]), hex(want[bad[0]]))


def _compile(tmp_path, prog, name):
    cp = compile_program(prog, str(tmp_path), name)
    return cp, rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)


def _oracle(cp, inputs_by_slot):
    fc = cp.flat
    return eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inputs_by_slot)


def test_multiplier2_docs_vector_and_random(tmp_path):
    cp, c = _compile(tmp_path, Program(Multiplier2()), "multiplier2")
    b = c.batch(1)
    b.set_inputs_json(0, '{"a": "3", "b": "11"}')
    b.run(); b.check_r1cs(); b.sync()
    assert b.status()[0] == 0 and b.witness(0) == [1, 33, 3, 11]
    p = tmp_path / "out.wtns"
    b.write_wtns(0, p)
    assert p.read_bytes() == wtns_bytes(c.q, [1, 33, 3, 11])
    assert len(p.read_bytes()) == 204
    b.close()
    rng = np.random.default_rng(0)
    n = 1000
    ins = [[int.from_bytes(rng.bytes(32), "little") % c.q for _ in range(2)] for _ in range(n)]
    b = c.batch(n)
    b.set_inputs(ins)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i in range(0, n, 37):
        assert b.witness(i) == [1, ins[i][0] * ins[i][1] % c.q, ins[i][0], ins[i][1]]
    b.close(); c.close()


def test_basic_main_golden_and_json_forms(tmp_path):
    cp, c = _compile(tmp_path, Program(BasicMain()), "basic")
    b = c.batch(3)
    b.set_inputs_json(0, '{"in": ["5", "7"]}')
    b.set_inputs_json(1, '{"in": ["0x10", "0b101"]}')
    b.set_input_signal(2, "in", 1, 9)
    b.set_input_signal(2, "in", 0, c.q - 1)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i, (x, y) in enumerate([(5, 7), (16, 5), (c.q - 1, 9)]):
        want, failed = _oracle(cp, {2: x, 3: y})
        assert failed is None and b.witness(i) == want
    b.close(); c.close()


def test_poseidon2_batch_bit_exact(tmp_path):
    cp, c = _compile(tmp_path, Program(Poseidon(2)), "poseidon2")
    rng = np.random.default_rng(1)
    n = 4096 + 37          # ragged: not a multiple of the block size
    ins = [[int.from_bytes(rng.bytes(32), "little") % c.q for _ in range(2)] for _ in range(n)]
    ins[0] = [1, 2]
    ins[1] = [0, 0]
    ins[2] = [c.q - 1, c.q - 1]
    b = c.batch(n)
    b.set_inputs(ins)
    with pytest.raises(rt.CwError, match="timing is off"):
        b.kernel_ms()
    b.set_timing(True)                                             # events around the parts of run / check on the batch's stream
    b.run()
    km = b.kernel_ms()
    assert km["ingest"] > 0 and km["eval"] > 0 and km["check"] is None        # the check has not run yet
    b.check_r1cs(); b.sync()
    km = b.kernel_ms()
    assert 0 < km["ingest"] < 50 and 0 < km["eval"] < 500 and 0 < km["check"] < 500
    b.set_timing(False)
    assert (b.status() == 0).all()
    assert b.signal(0, 1) == 7853200120776062878684798364095072458815029376092732009249414926327459813530
    # every instance: hash output vs the plain-integer Poseidon
    for i in range(0, n, 16):
        assert b.signal(i, 1) == poseidon_hash(c.q, ins[i]), i
    # sampled instances: the whole witness, byte for byte
    for i in (0, 1, 2, 255, 256, 4095, 4096, n - 1):
        want, failed = _oracle(cp, {2: ins[i][0], 3: ins[i][1]})
        assert failed is None
        assert b.witness_bytes(i) == b"".join(v.to_bytes(32, "little") for v in want), i
        assert check_r1cs(c.q, cp.flat.constraints, want) is None
    b.close(); c.close()


def test_asserts_and_slow_path_ops(tmp_path):
    # Num2Bits(8): in < 256 passes, in >= 256 trips `lc1 === in`; IsZero exercises select + div (INV)
    cp, c = _compile(tmp_path, Program(Num2Bits(8)), "n2b")
    vals = [0, 1, 255, 256, 1000, c.q - 1, 170]
    b = c.batch(len(vals))
    b.set_inputs([[v] for v in vals])
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    for i, v in enumerate(vals):
        want, failed = _oracle(cp, {cp.flat.main_input_start: v})
        if failed is None:
            assert st[i] == 0 and b.witness(i) == want
        else:
            assert st[i] & rt.ST_ASSERT_FAILED and st[i] & rt.ST_R1CS_FAILED, (v, st[i])
    b.close(); c.close()
    cp, c = _compile(tmp_path, Program(IsZero()), "iszero")
    vals = [0, 1, 2, c.q - 1, 12345678901234567890]
    b = c.batch(len(vals))
    b.set_inputs([[v] for v in vals])
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i, v in enumerate(vals):
        want, failed = _oracle(cp, {2: v})
        assert b.witness(i) == want and want[1] == int(v == 0)
    b.close(); c.close()


@pytest.mark.parametrize("lanes", ["16", "32", "64"])
@pytest.mark.parametrize("strands", ["1", "4", "16"])
def test_every_strand_count_and_lane_width_gives_the_same_witnesses(tmp_path, lanes, strands, monkeypatch):
    """cw_batch_create picks S and the instances per workgroup from the batch size; force every combination."""
    monkeypatch.setenv("CW_LANES", lanes)
    monkeypatch.setenv("CW_STRANDS", strands)
    cp, c = _compile(tmp_path, Program(Poseidon(2)), "poseidon2")
    B = 200
    rng = np.random.default_rng(int(lanes) + int(strands))
    ins = [[int.from_bytes(rng.bytes(32), "little") % c.q for _ in range(2)] for _ in range(B)]
    b = c.batch(B)
    assert (b.lanes, b.strands) == (int(lanes), int(strands))
    b.set_inputs(ins)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i in (0, 15, 16, 31, 32, 63, 64, 65, 199):
        assert b.signal(i, 1) == poseidon_hash(c.q, ins[i]), i
    want, failed = eval_flat(c.q, cp.flat.n_signals, cp.flat.n_temps, cp.flat.constants, cp.flat.code, {2: ins[77][0], 3: ins[77][1]})
    assert failed is None and b.witness(77) == want
    b.close(); c.close()


def test_bulk_witness_egress_equals_per_instance_egress(tmp_path):
    cp, c = _compile(tmp_path, Program(Poseidon(2)), "poseidon2")
    B = 333                                         # not a multiple of the 64-instance tile
    rng = np.random.default_rng(9)
    ins = [[int.from_bytes(rng.bytes(32), "little") % c.q for _ in range(2)] for _ in range(B)]
    b = c.batch(B)
    b.set_inputs(ins)
    b.run(); b.sync()
    allw = b.witnesses()
    assert allw.shape == (B, c.n_witness, 32)
    for i in (0, 1, 63, 64, 65, 200, B - 1):
        assert allw[i].tobytes() == b.witness_bytes(i), i
    part = b.witnesses(100, 70)
    assert part.tobytes() == allw[100:170].tobytes()
    # public signals = witness positions 1..n_public (Poseidon: the hash output)
    assert c.n_public == 1
    pub = b.public_signals()
    assert pub.shape == (B, 1, 32) and pub.tobytes() == allw[:, 1:2, :].tobytes()
    with pytest.raises(rt.CwError):
        b.witnesses(300, 40)
    b.close(); c.close()


@template
def BadMul(c):
    a = c.input("a")
    b = c.input("b")
    out = c.output("out")
    c.hint(out, a * b + 1)                      # witness code disagrees with the constraint
    c.enforce(out, a * b, runtime_check=False)  # constraint only (no run-time assert)


def test_r1cs_check_flags_a_corrupted_witness(tmp_path):
    cp, c = _compile(tmp_path, Program(BadMul()), "badmul")
    b = c.batch(300)
    b.set_inputs([[i + 1, i + 2] for i in range(300)])
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    assert ((st & rt.ST_R1CS_FAILED) != 0).all() and ((st & rt.ST_ASSERT_FAILED) == 0).all()
    assert (b.r1cs_first_bad() == 0).all()
    b.close(); c.close()


@template
def FlakyChain(c, n):
    # x[k+1] = x[k]^2 + b, but the witness code of link (a mod n) adds 1: every instance breaks a different row
    a = c.input("a")
    b = c.input("b")
    out = c.output("out")
    x = c.signal("x", n + 1)
    c.set(x[0], a + b)
    for k in range(n):
        c.hint(x[k + 1], x[k] * x[k] + b + (a % n).eq(k))
        c.enforce(x[k + 1], x[k] * x[k] + b, runtime_check=False)
    c.set(out, x[n] * 3 + x[1])


@pytest.mark.parametrize("mode", ["stream", "staged"])
def test_r1cs_check_reports_the_first_violated_row_per_instance(tmp_path, mode, monkeypatch):
    monkeypatch.setenv("CW_R1CS_MODE", mode)
    monkeypatch.setenv("CW_R1CS_CHUNKS", "5")          # several chunks: the failing row moves across them
    monkeypatch.setenv("CW_R1CS_TERMS", "40")
    n = 100
    cp, c = _compile(tmp_path, Program(FlakyChain(n)), "flaky")
    B = 300
    ins = [[i, 1000 + 7 * i] for i in range(B)]
    b = c.batch(B)
    b.set_inputs(ins)
    b.run(); b.check_r1cs(); b.sync()
    st, fb = b.status(), b.r1cs_first_bad()
    for i in range(B):
        w = b.witness(i)
        want = check_r1cs(c.q, cp.flat.constraints, w)
        assert want is not None
        assert (st[i] & rt.ST_R1CS_FAILED) and fb[i] == want, (i, fb[i], want)
    b.close(); c.close()


@template
def LooseBits(c, n):
    # out[k] is meant to be bit k of `a`, but the witness code of bit (b mod n) copies a whole 3-bit field instead: the boolean rows
    # out[k] * (out[k] - 1) = 0 (what Num2Bits writes per bit) are the only thing that can notice; written in all four shapes
    a = c.input("a")
    b = c.input("b")
    out = c.output("out", n)
    for k in range(n):
        wrong = (b % n).eq(k)
        c.hint(out[k], ((a >> k) & 1) + wrong * ((a >> k) & 6))
        if k % 4 == 0:
            c.enforce(out[k] * (out[k] - 1), 0, runtime_check=False)
        elif k % 4 == 1:
            c.enforce((out[k] - 1) * out[k], 0, runtime_check=False)
        elif k % 4 == 2:
            c.enforce(out[k] * (1 - out[k]), 0, runtime_check=False)
        else:
            c.enforce((1 - out[k]) * out[k], 0, runtime_check=False)


@pytest.mark.parametrize("mont", [False, True])
def test_r1cs_check_boolean_rows(tmp_path, mont, monkeypatch):
    """b * (b - 1) = 0 rows are checked as "b is 0 or 1" (one wire read, cw_r1cs_plan.h T_BOOL) on canonical and on
    Montgomery-form tables: the first violated row of every instance is the oracle's, and the general path
    (CW_R1CS_NO_BOOL=1) reports the same"""
    monkeypatch.setenv("CW_MONT", "1" if mont else "0")
    monkeypatch.setenv("CW_R1CS_TERMS", "24")
    n = 37
    cp, c = _compile(tmp_path, Program(LooseBits(n)), "loosebits%d" % mont)
    assert c.montgomery == mont
    B = 200
    ins = [[(i * 2654435761) % (1 << 40), i] for i in range(B)]
    seen = []
    for general in (False, True):
        if general:
            monkeypatch.setenv("CW_R1CS_NO_BOOL", "1")
        b = c.batch(B)
        b.set_inputs(ins)
        b.run(); b.check_r1cs(); b.sync()
        st, fb = b.status(), b.r1cs_first_bad()
        n_bad = 0
        for i in range(B):
            w = b.witness(i)
            want = check_r1cs(c.q, cp.flat.constraints, w)
            assert bool(st[i] & rt.ST_R1CS_FAILED) == (want is not None), i
            if want is not None:
                assert fb[i] == want, (i, fb[i], want)
                n_bad += 1
        assert 0 < n_bad < B
        seen.append((st.tolist(), fb.tolist()))
        b.close()
    assert seen[0] == seen[1]
    c.close()


@template
def LooseNum2Bits(c, n):
    # Num2Bits with the sum row `sum 2^k out[k] = a` next to the boolean rows, and witness code that writes a 3-bit field into bit
    # (b mod 2n) when b mod 2n < n: the boolean row of that bit AND the sum fail in that instance, nothing fails in the others
    a = c.input("a")
    b = c.input("b")
    out = c.output("out", n)
    acc = c.const(0)
    for k in range(n):
        wrong = (b % (2 * n)).eq(k)
        if k == 0:                                            # b mod 2n = 2n - 1: bit 0 flipped - still a bit, only the sum notices
            flip = (b % (2 * n)).eq(2 * n - 1)
            c.hint(out[k], (((a >> k) & 1) + flip) % 2 + wrong * ((a >> k) & 6))
        else:
            c.hint(out[k], ((a >> k) & 1) + wrong * ((a >> k) & 6))
        c.enforce(out[k] * (out[k] - 1), 0, runtime_check=False)
        acc = acc + out[k] * (1 << k)
    c.enforce(acc, a, runtime_check=False)


@pytest.mark.parametrize("mont", [False, True])
def test_r1cs_check_boolean_rows_folded_into_their_sum(tmp_path, mont, monkeypatch):
    """the boolean row of a bit rides inside the sum that reads the bit (cw_r1cs_plan.h build_stream): one read per bit, `b ? c : 0`
    for the term when the whole wave holds bits, the product otherwise - verdicts and first violated rows are the oracle's, and
    equal to the unfolded (CW_R1CS_NO_FOLD=1) and the general (CW_R1CS_NO_BOOL=1) plans'"""
    monkeypatch.setenv("CW_MONT", "1" if mont else "0")
    monkeypatch.setenv("CW_R1CS_TERMS", "24")
    n = 37
    cp, c = _compile(tmp_path, Program(LooseNum2Bits(n)), "loosen2b%d" % mont)
    assert c.montgomery == mont
    plan = c.r1cs_stream_plan(24)
    assert plan["n_folded"] == n and plan["n_bitsel"] == n - 1
    B = 333                                                     # waves 0-1 hold instances with a loose bit, wave 2 onwards only some
    # waves 0-2: loose bits in some lanes (the product form); waves 3-5: every lane holds bits (the select form), some sums fail
    ins = [[(i * 2654435761) % (1 << n), i if i < 192 else 2 * n * i + n + (i % n)] for i in range(B)]
    seen = []
    for env in (None, "CW_R1CS_NO_FOLD", "CW_R1CS_NO_BOOL"):
        if env:
            monkeypatch.setenv(env, "1")
        b = c.batch(B)
        b.set_inputs(ins)
        b.run(); b.check_r1cs(); b.sync()
        st, fb = b.status(), b.r1cs_first_bad()
        n_bad = 0
        for i in range(B):
            w = b.witness(i)
            want = check_r1cs(c.q, cp.flat.constraints, w)
            assert bool(st[i] & rt.ST_R1CS_FAILED) == (want is not None), (env, i)
            if want is not None:
                assert fb[i] == want, (env, i, fb[i], want)
                n_bad += 1
        assert 0 < n_bad < B
        seen.append((st.tolist(), fb.tolist()))
        b.close()
    assert seen[0] == seen[1] == seen[2]
    c.close()


def test_run_refuses_missing_inputs(tmp_path):
    cp, c = _compile(tmp_path, Program(Multiplier2()), "m2")
    b = c.batch(2)
    b.set_inputs_json(0, '{"a": 1, "b": 2}')
    with pytest.raises(rt.CwError) as e:
        b.run()
    assert "Not all inputs have been set" in str(e.value)
    b.close(); c.close()


@pytest.mark.parametrize("prime", ["bn128", "bls12381"])
def test_fp_mul_chain_matches_oracle(prime):
    f = Field(PRIMES[prime])
    rng = np.random.default_rng(5)
    n, iters = 4096, 64
    a = np.frombuffer(b"".join((int.from_bytes(rng.bytes(32), "little") % f.q).to_bytes(32, "little") for _ in range(n)), dtype=np.uint8).reshape(n, 32).copy()
    bb = np.frombuffer(b"".join((int.from_bytes(rng.bytes(32), "little") % f.q).to_bytes(32, "little") for _ in range(n)), dtype=np.uint8).reshape(n, 32).copy()
    out, ms = rt.fp_mul_bench(f.q, a, bb, iters)
    for i in range(0, n, 97):
        x = int.from_bytes(a[i].tobytes(), "little")
        y = int.from_bytes(bb[i].tobytes(), "little")
        for _ in range(iters):
            x = x * y * pow(1 << 261, -1, f.q) % f.q
        assert int.from_bytes(out[i].tobytes(), "little") == x


def test_process_level_cli_matches_reference_cli_contract(tmp_path):
    """`cw_witness <name> input.json out.wtns` = the reference's `./<name> input.json out.wtns` (main.cpp:336-373);
    a JSON array runs as one batch.  Golden bytes come from the reference runtime (tests/golden)."""
    import json
    import subprocess
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_wtns.json")))["cases"]
    assert rt.CLI_PATH.exists(), "cw_witness was not built (make -C circom_amd/csrc)"
    cp = compile_program(Program(Multiplier2()), str(tmp_path), "multiplier2")
    vec = gold["multiplier2"]["vectors"][0]
    (tmp_path / "in.json").write_text(json.dumps({"a": vec["inputs"][0], "b": vec["inputs"][1]}))
    r = subprocess.run([str(rt.CLI_PATH), str(tmp_path / "multiplier2"), str(tmp_path / "in.json"), str(tmp_path / "o.wtns")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "o.wtns").read_bytes().hex() == vec["wtns_hex"]
    # a batch: JSON array, one .wtns per instance; Num2Bits(16) rejects 65536 (instance 2) like the reference's assert
    cp = compile_program(Program(Num2Bits(16)), str(tmp_path), "num2bits16")
    vecs = gold["num2bits16"]["vectors"]
    rows = [{"in": vecs[0]["inputs"][0]}, {"in": vecs[2]["inputs"][0]}, {"in": "65536"}, {"in": vecs[3]["inputs"][0]}]
    (tmp_path / "many.json").write_text(json.dumps(rows))
    r = subprocess.run([str(rt.CLI_PATH), str(tmp_path / "num2bits16"), str(tmp_path / "many.json"), str(tmp_path / "w_%d.wtns")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "instance 2: Failed assert" in r.stderr
    for k, v in ((0, vecs[0]), (1, vecs[2]), (3, vecs[3])):
        assert (tmp_path / ("w_%d.wtns" % k)).read_bytes().hex() == v["wtns_hex"]
    assert not (tmp_path / "w_2.wtns").exists()


def test_bulk_wtns_files_and_failure_trace(tmp_path):
    """many .wtns files from one bulk transpose (byte-equal to the per-instance writer), and a readable
    trace of a failing instance with the .sym names of the violated constraint's wires."""
    from circom_amd.compiler import compile_program
    cp = compile_program(Program(FlakyChain(6)), str(tmp_path), "flaky6", sym=True)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    B = 70
    b = c.batch(B)
    b.set_inputs([[i, 100 + i] for i in range(B)])
    b.run(); b.check_r1cs(); b.sync()
    b.write_wtns_many(3, 66, str(tmp_path / "w_%u.wtns"))
    for i in (3, 40, 68):
        b.write_wtns(i, tmp_path / "one.wtns")
        assert (tmp_path / ("w_%d.wtns" % i)).read_bytes() == (tmp_path / "one.wtns").read_bytes()
    assert not (tmp_path / "w_2.wtns").exists() and not (tmp_path / "w_69.wtns").exists()
    with pytest.raises(rt.CwError):
        b.write_wtns_many(0, 2, str(tmp_path / "no_conversion.wtns"))
    fb = b.r1cs_first_bad()
    text = b.explain(5, cp.sym_path)
    assert "constraint %d" % fb[5] in text and "main.x[" in text and "violated" in text
    w = b.witness(5)
    k = int(fb[5])
    a_, b_, c_ = cp.flat.constraints[k]
    for sgn in list(a_) + list(b_) + list(c_):
        if sgn > 0:
            assert " = %d;" % w[sgn] in text
    assert "instance 5" in b.explain(5)                 # without a .sym: signal numbers
    b.close(); c.close()


@pytest.mark.gpu
def test_compact_container_of_a_256_bit_batch(tmp_path):
    """cw_write_wtnsb on the 256-bit engine (Poseidon): field elements, instance-major; expand(i) = the .wtns of instance i"""
    from circom_amd import runtime as rt, wtnsb
    from circom_amd.compiler import compile_program
    from circom_amd.frontend.dsl import Program
    from circom_amd.circuits.poseidon import Poseidon
    cp = compile_program(Program(Poseidon(2)), str(tmp_path), "poseidon2", sym=False, strands=(4,))
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    B = 100
    rng = np.random.default_rng(2)
    b = c.batch(B)
    b.set_inputs([[int.from_bytes(rng.bytes(31), "little") for _ in range(2)] for _ in range(B)])
    b.run(); b.sync()
    p = tmp_path / "p.wtnsb"
    b.write_wtnsb(p)
    w = wtnsb.load(p)
    assert (w.kind, w.batch, w.n_witness, w.prime) == (0, B, c.n_witness, c.q)
    for i in (0, 57, 99):
        q = tmp_path / ("i%d.wtns" % i)
        b.write_wtns(i, q)
        assert w.expand(i) == q.read_bytes()
    b.close(); c.close()
