"""The 256-bit schedule as EMITTED gfx950 code (hip_elements/fpjit.py + fpjit_bodies.py).

CPU: the row bodies compile within their register budgets; the IR of the emitted kernels, replayed with poisoned registers
and in-order memory queues (oracle/fpjit_eval.py), reproduces the schedule replay (oracle/tape_eval.py) on every operator,
on Poseidon in both value forms and on a strand-parallel EdDSA-style circuit; a wrong wait count is caught.
GPU: the emitted kernels against the interpreting kernel (same schedule, CW_FP_JIT=0) and against the oracle, bit for bit,
status words included."""
import os
import random

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.frontend.flatten import flatten
from circom_amd.circuits.basic import Multiplier2, IsZero, Num2Bits
from circom_amd.circuits.opzoo import OperatorZoo
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.hip_elements import fpjit, fpjit_bodies
from circom_amd.hip_elements.lower import lower
from oracle.field import PRIMES
from oracle.fpjit_eval import replay_tape, JitHazard
from oracle.tape_eval import eval_flat, eval_tape


@pytest.fixture(scope="module")
def bodies():
    return fpjit_bodies.build_bodies()


def test_bodies_respect_their_register_budgets(bodies):
    """parse_bodies() has already refused any body that touches a register outside its budget or contains a memory
    instruction; what is pinned here: the set exists in both parities, nothing but the heavy operators spills, and the
    product is the ~320 instructions the design counts on"""
    for base in ("add", "sub", "mmul", "mul2", "madd", "mulc0", "mulcp", "mulcn", "linp", "linn", "dotmac", "asserteq", "select", "ext"):
        assert base + "_e" in bodies and base + "_o" in bodies
    for name, b in bodies.items():
        assert b.text and b.n_instr > 0
        if b.parity not in ("h", "c", "k"):
            assert not b.scratch, name
        if b.parity == "e":
            assert not any(24 <= r < 40 for r in b.vwritten), name       # the odd rows' operands are in flight meanwhile
        if b.parity == "o":
            assert not any(r < 16 for r in b.vwritten), name
        if b.parity != "m":
            owned = (120, 121, 122, 125, 126, 127) + (() if b.chk else (123,))      # v123 = the fused check's finding
            assert not any(r in b.vwritten for r in owned) or b.parity in ("h", "c", "k"), name
    assert 250 <= bodies["mmul_e"].n_instr <= 400


def _inputs(fc, rng, q, small=False):
    return {fc.main_input_start + k: (rng.randrange(300) if small else rng.randrange(q)) for k in range(fc.n_main_inputs)}


@pytest.mark.parametrize("prime", ["bn128", "bls12381"])
def test_replay_of_emitted_ir_matches_schedule_replay_on_every_operator(bodies, prime):
    q = PRIMES[prime]
    fc = flatten(Program(OperatorZoo(), prime=prime))
    rng = random.Random(11)
    half = q >> 1
    cases = [(x, y) for x in (0, 1, 5, half, q - 1, 1 << 64) for y in (0, 1, 3, 255, q - 2)] + \
            [(rng.randrange(q), rng.randrange(q)) for _ in range(6)] + [(rng.randrange(q), rng.randrange(300)) for _ in range(6)]
    for S in (1, 4):
        t = lower(fc, n_strands=S)
        p = fpjit.emit(t, bodies)
        fpjit.assemble(p)
        assert p.code[:4] == b"\x7fELF"
        for x, y in cases:
            inp = {fc.main_input_start: x, fc.main_input_start + 1: y}
            want, st0 = eval_tape(t, inp)
            got, st1 = replay_tape(t, p, bodies, inp)
            assert st0 == st1 and got == want, (S, hex(x), hex(y))


@pytest.mark.parametrize("mont", [False, True])
@pytest.mark.parametrize("S", [1, 4, 16])
def test_replay_of_emitted_ir_poseidon(bodies, mont, S):
    fc = flatten(Program(Poseidon(2)))
    t = lower(fc, n_strands=S, mont=mont)
    p = fpjit.emit(t, bodies)
    rng = random.Random(S)
    for _ in range(2):
        inp = _inputs(fc, rng, fc.fp.q)
        want, st0 = eval_tape(t, inp)
        got, st1 = replay_tape(t, p, bodies, inp)
        assert (got, st1) == (want, st0)
    # the glue around a row is a few dozen instructions, not the interpreter's ~300
    assert p.stats["glue"] / p.stats["steps"] < 40


def test_spooled_text_assembles_to_the_same_code_object(bodies, tmp_path, monkeypatch):
    """programs of millions of rows write their text to a file as it is produced (the ECDSA verifier: ~80 M lines); forced
    here with a tiny threshold: the code object is the one the in-memory path assembles"""
    fc = flatten(Program(Poseidon(2)))
    t = lower(fc, n_strands=4, mont=True)
    a = fpjit.emit(t, bodies, fc.constraints)
    fpjit.assemble(a)
    orig = fpjit._Spool.__init__
    monkeypatch.setattr(fpjit._Spool, "__init__", lambda self, path=None, limit=0: orig(self, path, 997))
    b = fpjit.emit(t, bodies, fc.constraints, spool_path=str(tmp_path / "k.s"))
    assert b.asm is None and b.asm_path and b.ir == [[], [], [], []]
    fpjit.assemble(b)
    assert b.code == a.code and not (tmp_path / "k.s").exists()


def test_replay_catches_a_wait_that_is_too_weak(bodies):
    fc = flatten(Program(Poseidon(2)))
    t = lower(fc, n_strands=1, mont=True)
    p = fpjit.emit(t, bodies)
    inp = _inputs(fc, random.Random(1), fc.fp.q)
    ir = p.ir[0]
    k = next(i for i, ins in enumerate(ir) if ins[0] == "wait" and ins[1] is not None)
    saved = ir[k]
    ir[k] = ("wait", saved[1] - 2, saved[2])          # as if vmcnt were 2 larger: the operand's second half is still in flight
    with pytest.raises(JitHazard):
        replay_tape(t, p, bodies, inp)
    ir[k] = saved
    # and a body that clobbers a register group somebody still needs
    k = next(i for i, ins in enumerate(ir) if ins[0] == "call" and ins[1].startswith("mmul"))
    info = {n: (set(b.vwritten), b.parity) for n, b in bodies.items()}
    info[ir[k][1]] = (info[ir[k][1]][0] | set(range(0, 40)), info[ir[k][1]][1])
    from oracle.fpjit_eval import replay
    R = pow(2, t.rbits, t.q)
    with pytest.raises(JitHazard):
        replay(p.ir, info, t.q, t.n_signals, t.n_tslots, t.n_lds, {k2: v * R % t.q for k2, v in inp.items()}, t.rbits, R)


from circom_amd.frontend.dsl import template  # noqa: E402


@template
def Flaky(c, n):
    a = c.input("a")
    b = c.input("b")
    out = c.output("out")
    x = c.signal("x", n + 1)
    y = c.signal("y", n)
    c.set(x[0], a + b)
    for k in range(n):
        c.hint(x[k + 1], x[k] * x[k] + b + (a % (2 * n + 3)).eq(k))      # a product row breaks for a mod (2n + 3) == k
        c.enforce(x[k + 1], x[k] * x[k] + b, runtime_check=False)
        c.hint(y[k], x[k + 1] * 5 + x[k] * 7 + 3 + (a % (2 * n + 3)).eq(n + k))  # a linear row for a mod (2n + 3) == n + k
        c.enforce(y[k], x[k + 1] * 5 + x[k] * 7 + 3, runtime_check=False)
    c.set(out, x[n] * 3 + y[0])


def test_fused_check_finds_the_first_violated_row(bodies):
    """witness code that disagrees with its constraint (`<--` + `===` with the run-time assert off): the check steps the
    emitted code carries must name the row the oracle names, for every class of row they cover"""
    from oracle.tape_eval import check_r1cs
    from oracle import fpjit_eval

    n = 6
    fc = flatten(Program(Flaky(n)))
    q = fc.fp.q
    for S, mont in ((1, True), (4, False), (4, True)):
        t = lower(fc, n_strands=S, mont=mont)
        p = fpjit.emit(t, bodies, fc.constraints)
        assert sum(p.covered) >= len(fc.constraints) - 2 * n          # (a mod 2n).eq(k) rows have multi-term factors
        for a in range(2 * n + 3):
            inp = {fc.main_input_start: a, fc.main_input_start + 1: 1000 + a}
            sig, st = replay_tape(t, p, bodies, inp)
            want = None
            for ci, con in enumerate(fc.constraints):                 # the first violated row AMONG the fused ones
                if p.covered[ci] and check_r1cs(q, [con], sig) is not None:
                    want = ci
                    break
            assert fpjit_eval.replay.first_bad == want, (S, mont, a, fpjit_eval.replay.first_bad, want)
            assert (want is not None) == (a < 2 * n)


def test_replay_semaphore_style_strands(bodies):
    """EdDSA-style circuit (projective ladder hints): LINSUM rows of hundreds of terms, batched inversions, select / ext,
    LDS hand-offs between 16 strands"""
    from circom_amd.circuits import eddsa_host as H
    from circom_amd.circuits.eddsa import SemaphoreStyle
    q = PRIMES["bn128"]
    fc = flatten(Program(SemaphoreStyle(3, True)))
    row, _ = H.semaphore_inputs(q, 3, random.Random(5))
    inp = {fc.main_input_start + k: v for k, v in enumerate(row)}
    for S, mont in ((1, False), (16, True)):
        t = lower(fc, n_strands=S, mont=mont)
        p = fpjit.emit(t, bodies, fc.constraints)
        assert sum(p.covered) > 0.6 * len(fc.constraints)
        want, st0 = eval_tape(t, inp)
        got, st1 = replay_tape(t, p, bodies, inp)
        assert st0 == 0 and (got, st1) == (want, st0)
        from oracle import fpjit_eval
        assert fpjit_eval.replay.first_bad is None                    # a valid witness: no fused check fires
    bad = dict(inp)
    bad[fc.main_input_start + 2] = (row[2] + 1) % q           # tampered signature: the same status word
    want, st0 = eval_tape(t, bad)
    got, st1 = replay_tape(t, p, bodies, bad)
    assert st0 != 0 and st1 == st0


# ---- GPU ---------------------------------------------------------------------------------------------------------------
def _run_both(cp, rows, monkeypatch, strands, lanes=None, fused=None):
    """witness tables + status words of the same batch through the emitted code and through the interpreting kernel"""
    from circom_amd import runtime as rt
    out = []
    for emitted in (True, False):
        monkeypatch.setenv("CW_FP_JIT", "1" if emitted else "0")
        monkeypatch.setenv("CW_STRANDS", str(strands))
        if lanes:
            monkeypatch.setenv("CW_LANES", str(lanes))
        else:
            monkeypatch.delenv("CW_LANES", raising=False)
        if fused is None:
            monkeypatch.delenv("CW_FP_FUSED", raising=False)
        else:
            monkeypatch.setenv("CW_FP_FUSED", "1" if fused else "0")
        c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
        b = c.batch(len(rows))
        assert b.emitted == emitted and b.strands == strands
        if emitted and fused is not None and c.n_constraints:
            assert b.fused_check == fused
        b.set_inputs(rows)
        b.run()
        if c.n_constraints:
            b.check_r1cs()
        b.sync()
        out.append((b.witnesses().copy(), b.status().copy(), b.r1cs_first_bad().copy() if c.n_constraints else None))
        b.close(); c.close()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("prime", ["bn128", "bls12381", "secq256r1"])
def test_gpu_emitted_code_every_operator(tmp_path, monkeypatch, prime):
    from test_opzoo import _operands
    q = PRIMES[prime]
    cp = compile_program(Program(OperatorZoo(), prime=prime), str(tmp_path), "opzoo", sym=False, fpjit=True)
    assert {p.n_strands for p in cp.fpjit} == {1, 4, 16} and not any(p.covered for p in cp.fpjit)     # (no constraints)
    fc = cp.flat
    rows = _operands(q, 300, 9)
    for strands, lanes in ((1, None), (4, 32), (16, 16)):
        (w1, s1, _), (w0, s0, _) = _run_both(cp, rows, monkeypatch, strands, lanes)
        assert (s1 == s0).all() and (s1 == 0).all()
        assert w1.tobytes() == w0.tobytes(), (prime, strands)
    for i in range(0, len(rows), 37):
        inp = {fc.main_input_start: rows[i][0], fc.main_input_start + 1: rows[i][1]}
        sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None and w1[i].tobytes() == b"".join(v.to_bytes(32, "little") for v in sig)


@pytest.mark.gpu
@pytest.mark.parametrize("mont", [True, False])
def test_gpu_emitted_code_poseidon(tmp_path, monkeypatch, mont):
    from circom_amd.circuits.poseidon_constants import poseidon_hash
    cp = compile_program(Program(Poseidon(2)), str(tmp_path), "poseidon2", sym=False, mont=mont, fpjit=True)
    q = cp.flat.fp.q
    rng = random.Random(3)
    for n, strands, lanes, fused in ((700, 1, None, True), (333, 4, None, False), (100, 16, 16, True), (64 * 70, 4, None, None)):
        rows = [[rng.randrange(q), rng.randrange(q)] for _ in range(n)]
        (w1, s1, f1), (w0, s0, f0) = _run_both(cp, rows, monkeypatch, strands, lanes, fused)
        assert (s1 == 0).all() and (s0 == 0).all() and (f1 == f0).all()
        assert w1.tobytes() == w0.tobytes(), (n, strands)
        for i in (0, n // 2, n - 1):
            assert int.from_bytes(w1[i][1].tobytes(), "little") == poseidon_hash(q, rows[i])


@pytest.mark.gpu
def test_gpu_emitted_code_semaphore_style_and_status_words(tmp_path, monkeypatch):
    """the EdDSA-style circuit of BASELINE config 4 (projective hints) at a small tree depth: long LINSUM rows, batched
    inversions (the heavy body with its parked status word), LDS hand-offs; every fourth instance carries a tampered
    signature and must report the same first failing operation as the interpreter"""
    from circom_amd.circuits import eddsa_host as H
    from circom_amd.circuits.eddsa import SemaphoreStyle, SUBGROUP_ORDER
    q = PRIMES["bn128"]
    cp = compile_program(Program(SemaphoreStyle(4, True)), str(tmp_path), "sem4p", sym=False, fpjit=True)
    fc = cp.flat
    rng = random.Random(8)
    rows = []
    for i in range(150):
        row, _ = H.semaphore_inputs(q, 4, rng)
        row = list(row)
        if i % 4 == 3:
            row[2] = (row[2] + 1) % SUBGROUP_ORDER
        rows.append(row)
    for strands, lanes, fused in ((16, 16, True), (16, 16, False), (4, None, True), (1, None, True)):
        (w1, s1, f1), (w0, s0, f0) = _run_both(cp, rows, monkeypatch, strands, lanes, fused)
        assert (s1 == s0).all() and (f1 == f0).all(), strands
        assert all((s1[i] != 0) == (i % 4 == 3) for i in range(len(rows)))
        ok = [i for i in range(len(rows)) if i % 4 != 3]
        assert w1[ok].tobytes() == w0[ok].tobytes(), strands
    sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start + k: v for k, v in enumerate(rows[0])})
    assert failed is None and w1[0].tobytes() == b"".join(v.to_bytes(32, "little") for v in sig)


@pytest.mark.gpu
@pytest.mark.parametrize("mont", [True, False])
def test_gpu_fused_check_names_the_first_violated_row(tmp_path, monkeypatch, mont):
    """the check steps inside the emitted code + the stand-alone kernel on the rows they leave to it = the stand-alone
    kernel on every row (interpreter run): same status words, same first violated row, for every instance"""
    from oracle.tape_eval import check_r1cs
    n = 40
    cp = compile_program(Program(Flaky(n)), str(tmp_path), "flaky", sym=False, mont=mont, fpjit=True)
    assert all(sum(p.covered) >= len(cp.flat.constraints) - 2 * n for p in cp.fpjit if p.covered)
    assert sorted((p.n_strands, bool(p.covered)) for p in cp.fpjit) == [(1, False), (1, True), (4, False), (4, True), (16, False), (16, True)]
    rows = [[a, 1000 + 7 * a] for a in range(300)]
    for strands, lanes, fused in ((1, None, True), (4, None, True), (16, 16, True), (4, None, False)):
        (w1, s1, f1), (w0, s0, f0) = _run_both(cp, rows, monkeypatch, strands, lanes, fused)
        assert w1.tobytes() == w0.tobytes()
        assert (s1 == s0).all() and (f1 == f0).all(), (strands, [(i, f1[i], f0[i]) for i in range(300) if f1[i] != f0[i]][:5])
    from circom_amd import runtime as rt
    q = cp.flat.fp.q
    for i in (0, 5, n + 3, 2 * n + 1, 2 * n + 2, 299):
        sig = [int.from_bytes(w1[i][k].tobytes(), "little") for k in range(w1.shape[1])]
        want = check_r1cs(q, cp.flat.constraints, sig)
        assert (want is None) == ((s1[i] & rt.ST_R1CS_FAILED) == 0)
        if want is not None:
            assert f1[i] == want
    # CW_R1CS_AUDIT: the stand-alone kernel re-checks every row whatever the code covered - same words
    monkeypatch.setenv("CW_R1CS_AUDIT", "1")
    (w2, s2, f2), _ = _run_both(cp, rows, monkeypatch, 4, None, True)
    assert (s2 == s1).all() and (f2 == f1).all()


@pytest.mark.gpu
def test_gpu_emitted_code_runs_circom_functions(tmp_path, monkeypatch):
    """tier 2 inside the emitted code: BigMultModP's witness comes from long_div / short_div (value-dependent loops and
    branches, run-time indexed arrays) - the interpreter of csrc/cw_call.hip.h as one body of the straight-line program;
    every instance takes its own path through the functions"""
    from circom_amd.circuits.bigint import BigMultModP
    n, k = 32, 3
    # (one strand: schedules with calls on several strands run on the interpreting kernel - tests/test_functions.py)
    cp = compile_program(Program(BigMultModP(n, k), prime="bls12381"), str(tmp_path), "bigmultmodp", sym=False, strands=(1,), fpjit=True)
    assert cp.fpjit and all(p.n_strands == 1 for p in cp.fpjit)
    rnd = random.Random(12)
    rows = []
    for _ in range(500):
        p = rnd.randrange(1 << (n * k - 1), 1 << (n * k))
        a, b = rnd.randrange(p), rnd.randrange(p)
        rows.append([(x >> (n * i)) & ((1 << n) - 1) for x in (a, b, p) for i in range(k)])
    for fused in (False, True):
        (w1, s1, f1), (w0, s0, f0) = _run_both(cp, rows, monkeypatch, 1, None, fused)
        assert (s1 == 0).all() and (s0 == 0).all() and (f1 == f0).all()
        assert w1.tobytes() == w0.tobytes()
    fc = cp.flat
    for i in (0, 17, 499):
        sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code,
                                {fc.main_input_start + j: v for j, v in enumerate(rows[i])}, functions=fc.functions)
        assert failed is None and w1[i].tobytes() == b"".join(v.to_bytes(32, "little") for v in sig)


@pytest.mark.parametrize("seed", range(12))
def test_random_circuits_through_the_emitted_ir(bodies, seed):
    """the random circuits of tests/test_schedule_fuzz.py (long small-coefficient sums, field-sized coefficients, bit
    extraction, batched inversions, selects, wide fan-out, sub-components firing late) through the emitter: the replay of the
    emitted IR - rows and fused check, with poisoned registers and counted waits - reproduces the flat semantics for 1, 4
    and 16 strands, canonical and Montgomery-form tables, and no check step fires on a valid witness"""
    from test_schedule_fuzz import _random_template, Q
    from oracle import fpjit_eval
    rng = random.Random(2000 + seed)
    fc = flatten(Program(_random_template(seed, 40 + 26 * (seed % 11))))
    for S, mont in ((1, False), (4, True), (16, False), (16, True)):
        t = lower(fc, n_strands=S, mont=mont)
        p = fpjit.emit(t, bodies, fc.constraints)
        for trial in range(3):
            row = ([rng.randrange(Q) for _ in range(4)] if trial == 0 else [rng.randrange(4) for _ in range(4)] if trial == 1
                   else [Q - 1 - rng.randrange(3), 0, rng.randrange(Q), 1])
            inp = {fc.main_input_start + k: v for k, v in enumerate(row)}
            sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
            assert failed is None
            got, st = replay_tape(t, p, bodies, inp)
            assert st == 0 and got == sig, (seed, S, mont, trial)
            assert fpjit_eval.replay.first_bad is None, (seed, S, mont, trial, fpjit_eval.replay.first_bad)
    if seed < 3:
        fpjit.assemble(p)                          # the text is valid gfx950 assembly


def test_replay_of_emitted_ir_with_circom_functions(bodies):
    """tier 2 in the emitted code (CPU side): D_CALL steps of BigMultModP's long_div / short_div replayed through the IR -
    the call reads its arguments from and leaves its results in the register window of the value table - and an integer
    division by zero inside a function reaches the status word with the flat operation's index"""
    from circom_amd.circuits.bigint import BigMultModP
    from oracle import fpjit_eval
    n, k = 16, 2
    fc = flatten(Program(BigMultModP(n, k), prime="bls12381"))
    t = lower(fc, n_strands=1)
    p = fpjit.emit(t, bodies, fc.constraints)
    assert any(ins[0] == "callfn" for ins in p.ir[0]) and p.n_vgpr > 256          # the interpreter body's register budget
    fpjit.assemble(p)
    rnd = random.Random(4)
    for it in range(6):
        pp = rnd.randrange(1 << (n * k - 1), 1 << (n * k))
        a, b = rnd.randrange(pp), rnd.randrange(pp)
        vals = [(x >> (n * i)) & ((1 << n) - 1) for x in (a, b, pp) for i in range(k)]
        inp = {fc.main_input_start + j: v for j, v in enumerate(vals)}
        want, st0 = eval_tape(t, inp)
        got, st1 = replay_tape(t, p, bodies, inp)
        assert st0 == 0 and (got, st1) == (want, st0) and fpjit_eval.replay.first_bad is None
    # p = 0: long_div divides by zero inside the function
    inp = {fc.main_input_start + j: v for j, v in enumerate([5, 0, 7, 0, 0, 0])}
    want, st0 = eval_tape(t, inp)
    got, st1 = replay_tape(t, p, bodies, inp)
    assert st0 != 0 and st1 == st0


def test_replay_of_emitted_ir_library_circuits(bodies):
    """further shapes through the emitter and the IR replay: batched inversions with zero denominators (the reference's
    inv(0) = 0 inside a Montgomery-trick batch), a Merkle path of switchers and Poseidons, the affine BabyJubjub ladder
    (one inversion per step ON the chain: the heavy body with its parked registers, hundreds of times)"""
    from test_more_circuits import ThreeDivs, _merkle_case
    from circom_amd.circuits.merkle import MerkleTreeInclusionProof
    from circom_amd.circuits.babyjub import ScalarMulBits, BASE8
    from oracle import fpjit_eval
    q = PRIMES["bn128"]
    rng = random.Random(17)
    # (1) zero denominators in a batch of inversions
    fc = flatten(Program(ThreeDivs()))
    for S, mont in ((1, False), (4, True)):
        t = lower(fc, n_strands=S, mont=mont)
        p = fpjit.emit(t, bodies, fc.constraints)
        for zeros in range(8):
            row = [rng.randrange(q) for _ in range(3)] + [5 if (zeros >> i) & 1 else rng.randrange(q) for i in range(3)]
            inp = {fc.main_input_start + k: v for k, v in enumerate(row)}
            want, st0 = eval_tape(t, inp)
            got, st1 = replay_tape(t, p, bodies, inp)
            assert (got, st1) == (want, st0), (S, mont, zeros)
    # (2) Merkle path of depth 6
    fc = flatten(Program(MerkleTreeInclusionProof(6)))
    leaf, idx, sib, root = _merkle_case(q, 6, rng)
    row = [leaf] + idx + sib
    inp = {fc.main_input_start + k: v for k, v in enumerate(row)}
    for S in (1, 16):
        t = lower(fc, n_strands=S, mont=True)
        p = fpjit.emit(t, bodies, fc.constraints)
        got, st = replay_tape(t, p, bodies, inp)
        assert st == 0 and got[1] == root and fpjit_eval.replay.first_bad is None
        assert sum(p.covered) > 0.8 * len(fc.constraints)          # (the switchers' `s * (1 - s) = 0` has a two-term factor)
    # (3) affine ladder, 12 bits
    fc = flatten(Program(ScalarMulBits(12)))
    k = 0xB2D
    row = [(k >> i) & 1 for i in range(12)] + [BASE8[0], BASE8[1]]
    inp = {fc.main_input_start + j: v for j, v in enumerate(row)}
    t = lower(fc, n_strands=4, mont=True)
    p = fpjit.emit(t, bodies, fc.constraints)
    want, st0 = eval_tape(t, inp)
    got, st1 = replay_tape(t, p, bodies, inp)
    assert st0 == 0 and (got, st1) == (want, st0) and fpjit_eval.replay.first_bad is None


def test_no_assembler_on_the_host_leaves_a_valid_tape(tmp_path, monkeypatch):
    """`auto` emission must not fail the compile when clang / ld.lld of the ROCm LLVM are missing or reject the text -
    the tape then carries the interpreted program only (with a warning); an explicit fpjit=True still raises"""
    from circom_amd.hip_elements import bitjit
    from circom_amd import runtime as rt

    def boom():
        raise RuntimeError("no ROCm LLVM on this host")
    monkeypatch.setattr(bitjit, "_llvm_bin", boom)
    monkeypatch.delenv("CW_FPJIT", raising=False)
    with pytest.warns(UserWarning, match="not emitted"):
        cp = compile_program(Program(Poseidon(2)), str(tmp_path), "p2", sym=False, strands=(4,), fpjit="auto")
    assert cp.fpjit == ()
    rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path).close()
    with pytest.raises(RuntimeError, match="no ROCm LLVM"):
        compile_program(Program(Poseidon(2)), str(tmp_path), "p2b", sym=False, strands=(4,), fpjit=True)
    import glob
    import tempfile
    assert not glob.glob(os.path.join(tempfile.gettempdir(), "cw_fpjit_*", "k.s"))


# ---- round 6: several strands around run-time function calls, D_BITS rows ------------------------------------------------------------
def test_replay_of_emitted_ir_with_calls_on_several_strands(bodies):
    """Schedules of several strands whose rows include D_CALL (a heavy unit between FULL barriers) and D_BITS (one row = the
    stores of a whole Num2Bits) have an emitted form since round 6: the interpreter body `call_k` (compiled for the strand
    kernels' 128 VGPRs, spills in a private segment, native long_div included) and a `bits` step.  Replay == schedule replay."""
    from circom_amd.circuits.bigint import BigMultModP
    from circom_amd.hip_elements.lower import D_BITS, D_CALL
    assert bodies["call_k"].scratch_bytes > 0 and bodies["call_k"].parity == "k"
    n, k = 16, 2
    fc = flatten(Program(BigMultModP(n, k), prime="bls12381"))
    for S in (4, 16):
        t = lower(fc, n_strands=S)
        ops = np.asarray(t.rows)[:, 0] & 0xFF
        assert (ops == D_CALL).any() and (ops == D_BITS).any()
        p = fpjit.emit(t, bodies, None)
        assert p.n_vgpr == 128 and p.scratch_bytes == bodies["call_k"].scratch_bytes
        assert any(ins[0] == "bits" for strand in p.ir for ins in strand) and any(ins[0] == "callfn" for strand in p.ir for ins in strand)
        fpjit.assemble(p)
        rnd = random.Random(40 + S)
        for it in range(5):
            pp = rnd.randrange(1 << (n * k - 1), 1 << (n * k))
            a, b = rnd.randrange(pp), rnd.randrange(pp)
            vals = [(x >> (n * i)) & ((1 << n) - 1) for x in (a, b, pp) for i in range(k)]
            inp = {fc.main_input_start + j: v for j, v in enumerate(vals)}
            want, st0 = eval_tape(t, inp)
            got, st1 = replay_tape(t, p, bodies, inp)
            assert st0 == 0 and (got, st1) == (want, st0)


@pytest.mark.gpu
@pytest.mark.parametrize("S", [4, 16])
def test_gpu_emitted_code_with_calls_on_several_strands(tmp_path, monkeypatch, S):
    """the same on the device: BigMultModP on 4 / 16 strands through the emitted code (call_k body, bits steps) == the
    interpreting kernel == the oracle, every instance on its own path through long_div"""
    from circom_amd.circuits.bigint import BigMultModP
    n, k = 32, 3
    cp = compile_program(Program(BigMultModP(n, k), prime="bls12381"), str(tmp_path), "bigmultmodp_s%d" % S, sym=False, strands=(S,), fpjit=True)
    assert cp.fpjit and all(p.n_strands == S for p in cp.fpjit)
    rnd = random.Random(12)
    rows = []
    for _ in range(300):
        p = rnd.randrange(1 << (n * k - 1), 1 << (n * k))
        a, b = rnd.randrange(p), rnd.randrange(p)
        rows.append([(x >> (n * i)) & ((1 << n) - 1) for x in (a, b, p) for i in range(k)])
    (w1, s1, f1), (w0, s0, f0) = _run_both(cp, rows, monkeypatch, S, None, False)
    assert (s1 == 0).all() and (s0 == 0).all() and (f1 == f0).all()
    assert w1.tobytes() == w0.tobytes()
    fc = cp.flat
    for i in (0, 17, 299):
        sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code,
                                {fc.main_input_start + j: v for j, v in enumerate(rows[i])}, functions=fc.functions)
        assert failed is None and w1[i].tobytes() == b"".join(v.to_bytes(32, "little") for v in sig)
