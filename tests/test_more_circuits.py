"""More circuits on the same path: the BLS12-381 scalar field (`--prime bls12381`, BASELINE config 5's prime) and
a depth-20 Poseidon Merkle inclusion proof (the Merkle half of BASELINE config 4)."""
import random

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements.lower import lower
from circom_amd.hip_elements.writers import wtns_bytes
from circom_amd.circuits.basic import Num2Bits, IsZero, Multiplier2
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.poseidon_constants import poseidon_hash
from circom_amd.circuits.merkle import MerkleTreeInclusionProof
from oracle import ref_build
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, eval_tape, check_r1cs


def _merkle_case(q, depth, rng):
    leaf = rng.randrange(q)
    idx = [rng.randrange(2) for _ in range(depth)]
    sib = [rng.randrange(q) for _ in range(depth)]
    h = leaf
    for i in range(depth):
        h = poseidon_hash(q, [sib[i], h] if idx[i] else [h, sib[i]])
    return leaf, idx, sib, h


def test_merkle_depth20_root_r1cs_and_schedules():
    q = PRIMES["bn128"]
    fc = flatten(Program(MerkleTreeInclusionProof(20)))
    assert fc.inputs == [("leaf", 2, 1), ("pathIndices", 3, 20), ("siblings", 23, 20)]
    rng = random.Random(4)
    leaf, idx, sib, root = _merkle_case(q, 20, rng)
    inp = {2: leaf}
    inp.update({3 + i: b for i, b in enumerate(idx)})
    inp.update({23 + i: s for i, s in enumerate(sib)})
    sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    assert failed is None and sig[1] == root
    assert check_r1cs(q, fc.constraints, sig) is None
    for S in (1, 4):
        t = lower(fc, n_strands=S)
        got, st = eval_tape(t, inp)
        assert st == 0 and got == sig
    inp[3] = 2                                   # a non-boolean path index trips its `===`
    sig2, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    assert failed is not None


def test_bls12381_prime_reference_runtime_parity(tmp_path, ref_dir_bls12381):
    """`--prime bls12381`: same circuits over the BLS12-381 scalar field; the reference runtime rendered for that
    prime (oracle/_ref/bls12381) must produce the oracle's .wtns byte for byte."""
    q = PRIMES["bls12381"]
    rng = random.Random(12)
    for prog, name, cases in ((Program(Multiplier2(), prime="bls12381"), "multiplier2_bls", [[3, 11], [q - 1, q - 2]]),
                              (Program(Num2Bits(16), prime="bls12381"), "num2bits16_bls", [[0], [65535], [43690]]),
                              (Program(IsZero(), prime="bls12381"), "iszero_bls", [[0], [5], [q - 1]]),
                              (Program(Poseidon(2), prime="bls12381"), "poseidon2_bls",
                               [[1, 2], [rng.randrange(q), rng.randrange(q)]])):
        cp = compile_program(prog, str(tmp_path), name, sym=False, strands=(1, 4))
        fc = cp.flat
        assert fc.fp.q == q
        try:
            ref_build.build_circuit(cp)
        except RuntimeError as e:
            pytest.skip(str(e))
        raw = b"".join(v.to_bytes(32, "little") for row in cases for v in row)
        pre = str(tmp_path / (name + "_"))
        ref_build.run_loop(cp, raw, len(cases), 1, wtns_prefix=pre)
        for i, row in enumerate(cases):
            inp = {fc.main_input_start + k: v for k, v in enumerate(row)}
            want, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
            assert failed is None
            assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(q, want), (name, row)
            for t in cp_tapes(fc):
                got, st = eval_tape(t, inp)
                assert st == 0 and got == want
            if name == "poseidon2_bls":
                assert want[1] == poseidon_hash(q, row)


def cp_tapes(fc):
    return [lower(fc, n_strands=s) for s in (1, 4)]


@pytest.mark.gpu
def test_gpu_bls12381_and_merkle(tmp_path):
    from circom_amd import runtime as rt
    q = PRIMES["bls12381"]
    rng = random.Random(31)
    cp = compile_program(Program(Poseidon(2), prime="bls12381"), str(tmp_path), "poseidon2_bls", sym=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert c.q == q
    n = 300
    ins = [[rng.randrange(q), rng.randrange(q)] for _ in range(n)]
    b = c.batch(n)
    b.set_inputs(ins)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    fc = cp.flat
    for i in (0, 63, 64, 299):
        want, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {2: ins[i][0], 3: ins[i][1]})
        assert b.witness(i) == want
    b.close(); c.close()
    # Merkle depth 20, bn128, a batch with one bad path index
    q = PRIMES["bn128"]
    cp = compile_program(Program(MerkleTreeInclusionProof(20)), str(tmp_path), "merkle20", sym=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    n = 200
    cases = [_merkle_case(q, 20, rng) for _ in range(n)]
    rows = [[leaf] + idx + sib for leaf, idx, sib, _ in cases]
    rows[7][1] = 2
    b = c.batch(n)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    assert st[7] & rt.ST_ASSERT_FAILED and (np.delete(st, 7) == 0).all()
    for i in (0, 1, 100, 199):
        assert b.signal(i, 1) == cases[i][3]
    fc = cp.flat
    inp = {2 + k: v for k, v in enumerate(rows[5])}
    want, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    assert failed is None and b.witness(5) == want
    b.close(); c.close()


# ---- BabyJubjub scalar multiplication: drives DIV/INV (slow-path kernel variant) at scale ---------------------------
def _ed_add(p1, p2, q):
    from circom_amd.circuits.babyjub import A, D
    x1, y1 = p1
    x2, y2 = p2
    t = D * x1 * x2 * y1 * y2 % q
    return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, q) % q, (y1 * y2 - A * x1 * x2) * pow(1 - t, -1, q) % q)


def _ed_mul(k, p, q):
    acc = (0, 1)
    for i in range(k.bit_length() - 1, -1, -1):
        acc = _ed_add(acc, acc, q)
        if (k >> i) & 1:
            acc = _ed_add(acc, p, q)
    return acc


def test_babyjub_scalar_mul_vs_integer_arithmetic():
    from circom_amd.circuits.babyjub import ScalarMulBits, BASE8
    q = PRIMES["bn128"]
    n = 16
    fc = flatten(Program(ScalarMulBits(n)))
    rng = random.Random(8)
    for k in (0, 1, 2, 0xFFFF, rng.randrange(1 << n)):
        inp = {fc.main_input_start + i: (k >> i) & 1 for i in range(n)}
        inp[fc.main_input_start + n] = BASE8[0]
        inp[fc.main_input_start + n + 1] = BASE8[1]
        sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None and (sig[1], sig[2]) == _ed_mul(k, BASE8, q)
        assert check_r1cs(q, fc.constraints, sig) is None
        for S in (1, 4):
            t = lower(fc, n_strands=S)
            got, st = eval_tape(t, inp)
            assert st == 0 and got == sig
    # a point off the curve trips BabyCheck's `===`
    inp[fc.main_input_start + n] = 5
    sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    assert failed is not None


@pytest.mark.gpu
def test_gpu_babyjub_scalar_mul(tmp_path):
    from circom_amd import runtime as rt
    from circom_amd.circuits.babyjub import ScalarMulBits, BASE8
    q = PRIMES["bn128"]
    n = 32
    cp = compile_program(Program(ScalarMulBits(n)), str(tmp_path), "smul32", sym=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    rng = random.Random(5)
    B = 200
    ks = [rng.randrange(1 << n) for _ in range(B)]
    ks[0], ks[1] = 0, (1 << n) - 1
    rows = [[(k >> i) & 1 for i in range(n)] + list(BASE8) for k in ks]
    rows[9][n] = 7                                            # instance 9: point not on the curve
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    assert st[9] & rt.ST_ASSERT_FAILED and (np.delete(st, 9) == 0).all()
    for i in (0, 1, 2, 100, 199):
        assert (b.signal(i, 1), b.signal(i, 2)) == _ed_mul(ks[i], BASE8, q), i
    fc = cp.flat
    inp = {fc.main_input_start + k: v for k, v in enumerate(rows[3])}
    want, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    assert failed is None and b.witness(3) == want
    b.close(); c.close()


# ---- batched inversions (Montgomery's trick in the lowering): zero members keep the reference's inv(0) = 0 -------------
@template
def ThreeDivs(c):
    a = c.input("a", 3)
    b = c.input("b", 3)
    out = c.output("out", 3)
    chained = c.output("chained")
    for i in range(3):
        c.hint(out[i], a[i] / (b[i] - 5))                  # three independent divisions at one level: one batch
    c.hint(chained, (out[0] + 1) / (out[1] + out[2] + 2))  # depends on the batch: must be moved behind it


def test_batched_inversions_match_reference_semantics_for_zero_denominators(tmp_path, ref_dir_bn128):
    q = PRIMES["bn128"]
    fc = flatten(Program(ThreeDivs()))
    for S in (1, 4):
        t = lower(fc, n_strands=S)
        assert t.stats["inv_batches"] == 1 and t.stats["inv"] == 2      # 4 divisions -> 2 inversions
    rng = random.Random(12)
    cases = []
    for zeros in range(8):
        a = [rng.randrange(q) for _ in range(3)]
        b = [5 if (zeros >> i) & 1 else rng.randrange(q) for i in range(3)]
        cases.append(a + b)
    cases.append([1, 2, 3, 6, 6, 6])
    cases.append([0, 0, 0, 4, q - 1, 3])
    tapes = [lower(fc, n_strands=S) for S in (1, 4)]
    wants = []
    for row in cases:
        inp = {fc.main_input_start + k: v for k, v in enumerate(row)}
        sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None
        for i in range(3):
            d = (row[3 + i] - 5) % q
            assert sig[1 + i] == (row[i] * pow(d, -1, q) % q if d else 0)
        for t in tapes:
            got, st = eval_tape(t, inp)
            assert st == 0 and got == sig
        wants.append(sig)
    # the reference's own runtime agrees (Fr_div of a zero denominator gives 0, generic/fr.cpp:2895-2912)
    from oracle import ref_build
    from circom_amd.hip_elements.writers import wtns_bytes
    cp = compile_program(Program(ThreeDivs()), str(tmp_path), "threedivs", sym=False)
    try:
        ref_build.build_circuit(cp)
    except RuntimeError as e:
        pytest.skip(str(e))
    raw = b"".join(v.to_bytes(32, "little") for row in cases for v in row)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(cp, raw, len(cases), 1, wtns_prefix=pre)
    for i, want in enumerate(wants):
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(q, want), i


@pytest.mark.gpu
def test_gpu_batched_inversions_with_zero_denominators(tmp_path):
    from circom_amd import runtime as rt
    q = PRIMES["bn128"]
    cp = compile_program(Program(ThreeDivs()), str(tmp_path), "threedivs", sym=False)
    fc = cp.flat
    rng = random.Random(13)
    rows = []
    for k in range(256):
        a = [rng.randrange(q) for _ in range(3)]
        b = [5 if (k >> i) & 1 and k < 64 else rng.randrange(q) for i in range(3)]
        rows.append(a + b)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    b = c.batch(len(rows))
    b.set_inputs(rows)
    b.run(); b.sync()
    assert (b.status() == 0).all()
    for i in list(range(0, 16)) + [63, 64, 200, 255]:
        inp = {fc.main_input_start + k: v for k, v in enumerate(rows[i])}
        want, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None and b.witness(i) == want, i
    b.close(); c.close()


# ---- unusual component structures: firing order of the reference runtime ------------------------------------------------
@template
def _Const5(c):
    out = c.output("out")
    c.set(out, 5)

@template
def _AddK(c, k):
    a = c.input("a"); out = c.output("out")
    c.set(out, a + k)

@template
def _Pair(c):
    x = c.input("x", 2); out = c.output("out")
    inner = c.component("inner", _AddK(3))
    c.set(inner["a"], x[0] * x[1])
    c.set(out, inner["out"] + x[1])

@template
def _Odd(c):
    a = c.input("a"); b = c.input("b")
    out = c.output("out", 4)
    k5 = c.component("k5", _Const5())                    # no inputs: fires at creation
    cc = c.component("cc", _AddK(7))
    c.set(cc["a"], 11)                                  # constant input
    ps = [c.component("p", _Pair(), i) for i in range(3)]
    # inputs assigned out of order and interleaved
    c.set(ps[2]["x"][1], b)
    c.set(ps[0]["x"][0], a)
    c.set(ps[1]["x"][0], k5["out"])
    c.set(ps[2]["x"][0], cc["out"])
    c.set(ps[0]["x"][1], ps[2]["out"])
    c.set(ps[1]["x"][1], ps[0]["out"])
    c.set(out[0], ps[1]["out"])
    c.set(out[1], k5["out"] + cc["out"])
    c.set(out[2], ps[0]["out"] * ps[2]["out"])
    c.set(out[3], a)



def test_component_firing_order_matches_reference_runtime(tmp_path, ref_dir_bn128):
    """A component without inputs (fires at creation, template.rs:274-278), one fed by constants only, an array of
    components whose inputs are assigned out of order and interleaved (each fires when its LAST input arrives,
    store_bucket.rs:660-735), nesting: the flattened order must be the reference runtime's."""
    q = PRIMES["bn128"]
    cp = compile_program(Program(_Odd()), str(tmp_path), "oddshapes", sym=False, strands=(1,))
    fc = cp.flat
    try:
        ref_build.build_circuit(cp)
    except RuntimeError as e:
        pytest.skip(str(e))
    rng = random.Random(1)
    rows = [[3, 4], [0, 0], [q - 1, 2]] + [[rng.randrange(q), rng.randrange(q)] for _ in range(3)]
    raw = b"".join(v.to_bytes(32, "little") for r in rows for v in r)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=pre)
    t4 = lower(fc, n_strands=4)
    for i, r in enumerate(rows):
        inp = {fc.main_input_start + k: v for k, v in enumerate(r)}
        sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None and check_r1cs(q, fc.constraints, sig) is None
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(q, sig), i
        got, st = eval_tape(t4, inp)
        assert st == 0 and got == sig
    assert eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start: 3, fc.main_input_start + 1: 4})[0][1:5] == [1917, 23, 25201, 3]
