"""circomlib-structured BabyJubjub scalar multiplication (circuits/escalarmul.py: Montgomery-form ladders with offset,
3-bit windows behind MultiMux3 tables, segments) against plain-integer Edwards arithmetic, the R1CS, the lowered schedules
and - on the GPU - the device, plus the windowed EdDSA verifier / Semaphore-style relation built on it."""
import random

import pytest

from circom_amd.circuits import eddsa_host as H
from circom_amd.circuits.babyjub import BASE8
from circom_amd.circuits.escalarmul import (EscalarMulAny, EscalarMulFix, Edwards2Montgomery, Montgomery2Edwards, MontgomeryAdd,
                                            MontgomeryDouble, MultiMux3, WindowMulFix)
from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements.lower import lower
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, eval_tape, check_r1cs

Q = PRIMES["bn128"]


def _run(fc, values):
    inp = {fc.main_input_start + k: v % Q for k, v in enumerate(values)}
    sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    assert failed is None
    assert check_r1cs(Q, fc.constraints, sig) is None
    return sig, inp


def _bits(k, n):
    return [(k >> i) & 1 for i in range(n)]


def test_montgomery_form_round_trip_add_and_double():
    @template
    def T(c):
        p = c.input("p", 2); q2 = c.input("q", 2)
        out = c.output("out", 6)
        e1 = c.component("e1", Edwards2Montgomery()); e2 = c.component("e2", Edwards2Montgomery())
        for k in range(2):
            c.set(e1["in"][k], p[k]); c.set(e2["in"][k], q2[k])
        add = c.component("add", MontgomeryAdd()); dbl = c.component("dbl", MontgomeryDouble())
        for k in range(2):
            c.set(add["in1"][k], e1["out"][k]); c.set(add["in2"][k], e2["out"][k]); c.set(dbl["in"][k], e1["out"][k])
        b0 = c.component("b0", Montgomery2Edwards()); b1 = c.component("b1", Montgomery2Edwards()); b2 = c.component("b2", Montgomery2Edwards())
        for k in range(2):
            c.set(b0["in"][k], e1["out"][k]); c.set(b1["in"][k], add["out"][k]); c.set(b2["in"][k], dbl["out"][k])
        for k in range(2):
            c.set(out[k], b0["out"][k]); c.set(out[2 + k], b1["out"][k]); c.set(out[4 + k], b2["out"][k])

    fc = flatten(Program(T()))
    rng = random.Random(2)
    for _ in range(5):
        P = H.ed_mul(rng.randrange(1, 1 << 200), BASE8, Q)
        R = H.ed_mul(rng.randrange(1, 1 << 200), BASE8, Q)
        sig, _ = _run(fc, [P[0], P[1], R[0], R[1]])
        assert tuple(sig[1:3]) == P and tuple(sig[3:5]) == H.ed_add(P, R, Q) and tuple(sig[5:7]) == H.ed_add(P, P, Q)


def test_window_table_selects_every_multiple():
    @template
    def T(c):
        s = c.input("s", 3)
        out = c.output("out", 4)
        e = c.component("e", Edwards2Montgomery())
        c.set(e["in"][0], BASE8[0]); c.set(e["in"][1], BASE8[1])
        w = c.component("w", WindowMulFix())
        for j in range(3):
            c.set(w["in"][j], s[j])
        c.set(w["base"][0], e["out"][0]); c.set(w["base"][1], e["out"][1])
        b = c.component("b", Montgomery2Edwards()); b8 = c.component("b8", Montgomery2Edwards())
        for k in range(2):
            c.set(b["in"][k], w["out"][k]); c.set(b8["in"][k], w["out8"][k])
        for k in range(2):
            c.set(out[k], b["out"][k]); c.set(out[2 + k], b8["out"][k])

    fc = flatten(Program(T()))
    for v in range(8):
        sig, _ = _run(fc, _bits(v, 3))
        assert tuple(sig[1:3]) == H.ed_mul(v + 1, BASE8, Q) and tuple(sig[3:5]) == H.ed_mul(8, BASE8, Q)


@pytest.mark.parametrize("n", [4, 9, 253])
def test_escalarmulfix_matches_integer_arithmetic(n):
    fc = flatten(Program(EscalarMulFix(n, BASE8)))
    rng = random.Random(n)
    ks = [0, 1, 2, 7, 8, (1 << n) - 1, 1 << (n - 1)] + [rng.randrange(1 << n) for _ in range(3 if n > 100 else 6)]
    for k in ks:
        k %= 1 << n
        sig, inp = _run(fc, _bits(k, n))
        assert tuple(sig[1:3]) == H.ed_mul(k, BASE8, Q), k
    if n == 253:                     # two segments (246 + 7 bits); the lowered schedule agrees
        got, st = eval_tape(lower(fc, n_strands=4, mont=True), inp)
        assert st == 0 and got == sig


@pytest.mark.parametrize("n", [3, 10, 150, 254])
def test_escalarmulany_matches_integer_arithmetic(n):
    fc = flatten(Program(EscalarMulAny(n)))
    rng = random.Random(n)
    P = H.ed_mul(rng.randrange(1, 1 << 240), BASE8, Q)
    ks = [0, 1, 2, 3, (1 << n) - 1, 1 << (n - 1)] + [rng.randrange(1 << n) for _ in range(2 if n > 100 else 6)]
    for k in ks:
        sig, inp = _run(fc, _bits(k, n) + [P[0], P[1]])
        assert tuple(sig[1:3]) == H.ed_mul(k, P, Q), k
    # the identity in: the identity out, whatever the scalar (the ladder runs on BASE8, the result is masked)
    sig, _ = _run(fc, _bits(ks[-1], n) + [0, 1])
    assert tuple(sig[1:3]) == (0, 1)
    # a point of the full group (cofactor 8): the same relation
    sig, inp = _run(fc, _bits(ks[-1], n) + list(H.ed_mul(3, P, Q)))
    assert tuple(sig[1:3]) == H.ed_mul(ks[-1] * 3, P, Q)
    if n == 150:                     # two segments (148 + 2 bits); the lowered schedule agrees
        got, st = eval_tape(lower(fc, n_strands=4, mont=True), inp)
        assert st == 0 and got == sig


def test_windowed_eddsa_verifier_and_semaphore_relation():
    """the circomlib-structured EdDSA verifier inside the Semaphore-style relation: same inputs, same outputs as the
    bit-serial relation, every constraint satisfied; a tampered signature fails an `===`; the lowered strand schedule and the
    emitted code's IR reproduce the flat semantics"""
    from circom_amd.circuits.eddsa import SemaphoreStyle, SUBGROUP_ORDER
    levels = 3
    fcw = flatten(Program(SemaphoreStyle(levels, "window")))
    fcb = flatten(Program(SemaphoreStyle(levels, False)))
    rng = random.Random(21)
    row, (root, nullifier) = H.semaphore_inputs(Q, levels, rng)
    sig, inp = _run(fcw, row)
    assert (sig[1], sig[2]) == (root, nullifier)
    sigb, _ = _run(fcb, row)
    assert (sigb[1], sigb[2]) == (root, nullifier)
    # (at --O0 the component wiring of the Montgomery-form chain outweighs its cheaper steps: 46 823 constraints at depth 20
    # against 43 275 for the bit-serial ladder; what differs is the witness program - one division hint per step, not two)
    bad = list(row)
    bad[2] = (row[2] + 1) % SUBGROUP_ORDER
    _, failed = eval_flat(Q, fcw.n_signals, fcw.n_temps, fcw.constants, fcw.code, {fcw.main_input_start + k: v for k, v in enumerate(bad)})
    assert failed is not None
    t = lower(fcw, n_strands=16, mont=True)
    got, st = eval_tape(t, inp)
    assert st == 0 and got == sig
    from circom_amd.hip_elements import fpjit, fpjit_bodies
    from oracle import fpjit_eval
    bodies = fpjit_bodies.build_bodies()
    p = fpjit.emit(t, bodies, fcw.constraints)
    got2, st2 = fpjit_eval.replay_tape(t, p, bodies, inp)
    assert st2 == 0 and got2 == sig and fpjit_eval.replay.first_bad is None


@pytest.fixture(scope="module")
def semw(tmp_path_factory):
    from conftest import emit_for_gpu
    from circom_amd.compiler import compile_program
    from circom_amd.circuits.eddsa import SemaphoreStyle
    d = tmp_path_factory.mktemp("semw")
    return compile_program(Program(SemaphoreStyle(20, "window")), str(d), "semaphore20w", sym=False, strands=(16,), fpjit=emit_for_gpu())


def test_windowed_semaphore_reference_runtime_wtns_equal_oracle(semw, tmp_path, ref_dir_bn128):
    """BASELINE config 4's relation with circomlib's EdDSA structure (depth 20: 46 841 signals, 46 823 constraints at --O0):
    the reference's own C++ runtime executes it (through the .dat this repo writes) and writes the oracle's bytes"""
    from circom_amd.hip_elements.writers import wtns_bytes
    from oracle import ref_build
    try:
        ref_build.build_circuit(semw)
    except RuntimeError as e:
        pytest.skip(str(e))
    fc = semw.flat
    assert (fc.n_signals, len(fc.constraints)) == (46841, 46823)
    rng = random.Random(31)
    rows = [H.semaphore_inputs(Q, 20, rng)[0] for _ in range(3)]
    raw = b"".join(v.to_bytes(32, "little") for r in rows for v in r)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(semw, raw, len(rows), 1, wtns_prefix=pre)
    for i, r in enumerate(rows):
        want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start + k: v for k, v in enumerate(r)})
        assert failed is None and check_r1cs(Q, fc.constraints, want) is None
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(Q, want), i


@pytest.mark.gpu
def test_gpu_windowed_semaphore_matches_oracle_and_reference(semw, tmp_path):
    import numpy as np
    from circom_amd import runtime as rt
    from circom_amd.circuits.eddsa import SUBGROUP_ORDER
    from oracle import ref_build
    fc = semw.flat
    rng = random.Random(33)
    B = 96
    rows, outs = zip(*(H.semaphore_inputs(Q, 20, rng) for _ in range(B)))
    rows = [list(r) for r in rows]
    rows[5][2] = (rows[5][2] + 1) % SUBGROUP_ORDER          # instance 5: forged signature
    c = rt.Circuit(semw.tape_path, semw.dat_path, semw.r1cs_path)
    b = c.batch(B)
    assert b.emitted and b.strands == 16                     # the rows run as emitted code
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    assert st[5] & rt.ST_ASSERT_FAILED and (np.delete(st, [5]) == 0).all()
    for i in (0, 1, 50, 95):
        assert (b.signal(i, 1), b.signal(i, 2)) == outs[i], i
    want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start + k: v for k, v in enumerate(rows[3])})
    assert failed is None and b.witness(3) == want
    _, loop = ref_build.binaries("bn128", "semaphore20w")
    if loop.exists():                                       # prebuilt by __graft_entry__.build()
        idx = [k for k in range(B) if k != 5][:6]
        raw = b"".join(v.to_bytes(32, "little") for i in idx for v in rows[i])
        pre = str(tmp_path / "ref_")
        ref_build.run_loop(semw, raw, len(idx), 1, wtns_prefix=pre)
        for j, i in enumerate(idx):
            g = tmp_path / ("gpu_%d.wtns" % i)
            b.write_wtns(i, g)
            assert g.read_bytes() == open(pre + "%d.wtns" % j, "rb").read(), i
    b.close(); c.close()
