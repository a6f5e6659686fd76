"""The pipelined single-wave variant of the schedule (hip_elements/pipe.py, csrc cw_pipe_kernel): LDS result ring + load
lists issued one batch ahead, so that no row waits for the value table.

CPU: the planner's output replayed with the kernel's timing (oracle/tape_eval.py eval_pipe: loads read the table when the
previous batch starts and land when their batch starts, operands of the next row are read before the current row writes)
gives the reference's signals for every batch shape; the replay and the loader reject broken schedules.
GPU: the kernel's witnesses equal the oracle's and those of the strand variants, for every batch shape and lane width."""
import os
import random
import struct

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from circom_amd.circuits.basic import Num2Bits, IsZero
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.poseidon_constants import poseidon_hash
from circom_amd.circuits.stdlib import LessThan
from circom_amd.circuits.babyjub import ScalarMulBitsProj, ScalarMulBits, BASE8
from circom_amd.hip_elements import writers
from circom_amd.hip_elements.lower import lower
from circom_amd.hip_elements import pipe as PP
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, eval_tape, ScheduleHazard

Q = PRIMES["bn128"]
SHAPES = ((8, 8), (8, 4), (4, 4))


def _inp(fc, row):
    return {fc.main_input_start + k: v for k, v in enumerate(row)}


@template
def _Mix(c):
    """long sums (split into chains), a wide fan-out (more than two destinations), divisions, a select, comparisons"""
    x = c.input("x", 3)
    out = c.output("out", 4)
    bits = c.component("bits", Num2Bits(40))
    c.set(bits["in"], x[0])
    acc = c.const(7)
    for i in range(40):
        acc = acc + bits["out"][i] * (3 * i + 1)
    s = c.signal("s")
    c.set(s, acc)
    fan = c.signal("fan", 9)
    for i in range(9):
        c.set(fan[i], s)                               # nine copies of one value
    lt = c.component("lt", LessThan(41))
    c.set(lt["in"][0], x[0]); c.set(lt["in"][1], s)
    z = c.component("z", IsZero())
    c.set(z["in"], x[1] - x[2])
    d = c.signal("d")
    c.hint(d, (x[1] + 5) / (x[2] + 3))
    c.enforce(d * (x[2] + 3), x[1] + 5)
    c.set(out[0], fan[8] * fan[0] + lt["out"])
    c.set(out[1], d * d + z["out"])
    c.hint(out[2], c.select(x[1].lt(x[2]), d, s))
    c.set(out[3], x[0] * 0x1234567890ABCDEF1234567890ABCDEF1234567890 + x[1] * (Q - 5) + x[2] * (Q // 3) + d * 77)


def _cases():
    rng = random.Random(3)
    yield "poseidon2", flatten(Program(Poseidon(2))), [[rng.randrange(Q), rng.randrange(Q)] for _ in range(2)]
    yield "mix", flatten(Program(_Mix())), [[rng.randrange(1 << 40), rng.randrange(Q), rng.randrange(Q)],
                                           [5, 9, 9], [0, 0, Q - 4]]
    e = [rng.randrange(2) for _ in range(12)]
    yield "ladder", flatten(Program(ScalarMulBitsProj(12))), [e + list(BASE8)]


@pytest.mark.parametrize("shape", SHAPES)
def test_pipelined_schedule_replays_to_the_reference_signals(shape):
    for name, fc, rows in _cases():
        tp = lower(fc, pipe=shape)
        assert tp.kind == 1 and tp.pipe == (shape[0], shape[1], 2 * shape[0]) and len(tp.rows) % shape[0] == 0
        assert tp.stats["pipe_rows"] == len(tp.rows) and tp.n_lds == 2 * shape[0] + 2 * shape[1]
        for row in rows:
            want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, row))
            got, st = eval_tape(tp, _inp(fc, row))
            assert failed is None and st == 0 and got == want, (name, shape)


def test_failing_assert_and_projective_ladder_equals_affine_ladder():
    fc = flatten(Program(_Mix()))
    tp = lower(fc, pipe=(8, 8))
    # x[2] = -3: the denominator is 0, the reference's inverse gives 0 and `d * 0 === x1 + 5` fails
    want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, [1, 2, Q - 3]))
    got, st = eval_tape(tp, _inp(fc, [1, 2, Q - 3]))
    assert failed is not None and st & 0xFF == 1
    rng = random.Random(8)
    e = [rng.randrange(2) for _ in range(20)]
    fa, fp_ = flatten(Program(ScalarMulBits(20))), flatten(Program(ScalarMulBitsProj(20)))
    ra = eval_flat(Q, fa.n_signals, fa.n_temps, fa.constants, fa.code, _inp(fa, e + list(BASE8)))[0]
    rp, st = eval_tape(lower(fp_, pipe=(4, 4)), _inp(fp_, e + list(BASE8)))
    assert st == 0 and (ra[1], ra[2]) == (rp[1], rp[2])


def test_replay_rejects_broken_schedules():
    fc = flatten(Program(Poseidon(2)))
    inp = _inp(fc, [3, 4])
    tp = lower(fc, pipe=(8, 8))
    # an operand that reads a ring entry nobody has written yet
    bad = lower(fc, pipe=(8, 8))
    r = next(i for i in range(len(bad.rows)) if (bad.rows[i, 0] >> 8) & 7 == 4 and bad.rows[i, 2] & 0xFF >= 16)
    bad.rows[r, 2] = (int(bad.rows[r, 2]) & ~0xFF) | 15
    if r < 15:
        with pytest.raises(ScheduleHazard):
            eval_tape(bad, inp)
    else:
        assert eval_tape(bad, inp)[0] != eval_tape(tp, inp)[0]
    # a staged operand taken from the half that is being filled
    bad = lower(fc, pipe=(8, 8))
    r = next(i for i in range(len(bad.rows)) if (bad.rows[i, 0] >> 8) & 7 == 4 and bad.rows[i, 2] & 0xFF >= 16)
    e = int(bad.rows[r, 2]) & 0xFF
    bad.rows[r, 2] = (int(bad.rows[r, 2]) & ~0xFF) | (e + 8 if e < 24 else e - 8)
    with pytest.raises(ScheduleHazard):
        eval_tape(bad, inp)
    # a store target on a row that has no value
    bad = lower(fc, pipe=(8, 8))
    r = next(i for i in range(len(bad.rows)) if bad.rows[i, 0] & 0xFF == PP.D_NOP)
    bad.rows[r, 3] = 5
    with pytest.raises(ScheduleHazard):
        eval_tape(bad, inp)


def _write_only_pipe(tmp_path, fc, tp, name):
    p = lambda ext: os.path.join(str(tmp_path), name + ext)
    writers.write_tape(p(".cwt"), [tp])
    writers.write_dat(p(".dat"), fc)
    writers.write_r1cs(p(".r1cs"), fc)
    return p(".cwt"), p(".dat"), p(".r1cs")


def test_loader_accepts_the_variant_and_rejects_every_index_out_of_range(tmp_path):
    from circom_amd import runtime as rt
    fc = flatten(Program(Poseidon(2)))
    good = lower(fc, pipe=(8, 8))
    c = rt.Circuit(*_write_only_pipe(tmp_path, fc, good, "ok"))
    assert c.n_signals == fc.n_signals
    c.close()
    lds = next(i for i in range(len(good.rows)) if (good.rows[i, 0] >> 8) & 7 == 4)
    val = next(i for i in range(len(good.rows)) if good.rows[i, 3] != PP.P_NONE)
    nop = next(i for i in range(len(good.rows)) if good.rows[i, 0] & 0xFF == PP.D_NOP)
    dot = next(i for i in range(len(good.rows)) if good.rows[i, 0] & 0xFF == 36)

    def mutations():
        yield "entry", lambda t: t.rows.__setitem__((lds, 2), (int(t.rows[lds, 2]) & ~0xFF) | 40)
        def other_half(t):        # row 0 belongs to batch 0, whose loads land in staging entries 16..23
            t.rows[0, 0] = (int(t.rows[0, 0]) & ~(7 << 11)) | (4 << 11)
            t.rows[0, 2] = (int(t.rows[0, 2]) & ~0xFF00) | (28 << 8)
        yield "other half", other_half
        yield "store", lambda t: t.rows.__setitem__((val, 3), fc.n_signals)
        yield "store tmp", lambda t: t.rows.__setitem__((val, 4), (1 << 31) | t.n_tslots)
        yield "store on nop", lambda t: t.rows.__setitem__((nop, 4), 1)
        yield "ring entry", lambda t: t.rows.__setitem__((val, 2), (int(t.rows[val, 2]) & ~0xFF0000) | (((val + 1) % 16) << 16))
        yield "opcode", lambda t: t.rows.__setitem__((nop, 0), 29)                       # BARRIER
        yield "call", lambda t: t.rows.__setitem__((nop, 0), 37)
        yield "kind", lambda t: t.rows.__setitem__((lds, 0), int(t.rows[lds, 0]) | (7 << 8))
        yield "terms", lambda t: t.rows.__setitem__((dot, 1), 10 ** 6)
        yield "term entry", lambda t: t.terms.__setitem__((0, 1), 99)
        yield "term kind", lambda t: t.terms.__setitem__((0, 0), 0)
        yield "term const", lambda t: t.terms.__setitem__((0, 2), len(t.lconsts))
        yield "load const", lambda t: t.extras.__setitem__(0, (1 << 30) | len(t.consts))
        yield "load slot", lambda t: t.extras.__setitem__(1, fc.n_signals + 7)
        yield "load count", lambda t: setattr(t, "extras", t.extras[:-8])
        yield "partial batch", lambda t: setattr(t, "rows", t.rows[:-1])
        yield "shape", lambda t: setattr(t, "pipe", (8, 5, 16))
        yield "lds", lambda t: setattr(t, "n_lds", 31)

    for what, mutate in mutations():
        t = lower(fc, pipe=(8, 8))
        mutate(t)
        if what in ("load count", "partial batch"):
            t.stream_off = np.asarray([0, len(t.rows)], dtype=np.uint32)
            t.extra_off = np.asarray([0, len(t.extras)], dtype=np.uint32)
        with pytest.raises(rt.CwError):
            rt.Circuit(*_write_only_pipe(tmp_path, fc, t, "bad"))
            pytest.fail("the loader accepted a schedule with a broken " + what)


def test_compile_program_adds_the_variant_last_and_skips_it_for_bit_circuits_and_functions(tmp_path):
    cp = compile_program(Program(Poseidon(2)), str(tmp_path), "p2", sym=False, pipe=(8, 8))
    raw = open(cp.tape_path, "rb").read()
    assert struct.unpack_from("<I", raw, 12)[0] == 4                    # strands 1, 4, 16 + the pipelined variant
    cp = compile_program(Program(Poseidon(2)), str(tmp_path), "p2", sym=False)
    assert struct.unpack_from("<I", open(cp.tape_path, "rb").read(), 12)[0] == 3   # opt-in
    from circom_amd.circuits.bigint import BigMod
    cp = compile_program(Program(BigMod(16, 2)), str(tmp_path), "bm", sym=False, pipe=(8, 8))
    assert struct.unpack_from("<I", open(cp.tape_path, "rb").read(), 12)[0] == 3   # function calls: no pipelined variant
    with pytest.raises(ValueError):
        lower(cp.flat, pipe=(8, 8))


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("lanes", ["16", "64"])
def test_gpu_pipelined_kernel_matches_the_oracle(tmp_path, shape, lanes, monkeypatch):
    from circom_amd import runtime as rt
    monkeypatch.setenv("CW_LANES", lanes)
    rng = random.Random(shape[0] * 16 + shape[1])
    for name, fc, _ in _cases():
        tp = lower(fc, pipe=shape)
        c = rt.Circuit(*_write_only_pipe(tmp_path, fc, tp, name))
        B = 150
        if name == "ladder":
            rows = [[rng.randrange(2) for _ in range(12)] + list(BASE8) for _ in range(B)]
        elif name == "mix":
            rows = [[rng.randrange(1 << 40), rng.randrange(Q), rng.randrange(Q)] for _ in range(B)]
            rows[7] = [1, 2, Q - 3]                                   # division by zero: the `===` fails
        else:
            rows = [[rng.randrange(Q) for _ in range(fc.n_main_inputs)] for _ in range(B)]
        b = c.batch(B)
        assert b.pipelined == shape and b.lanes == int(lanes)
        b.set_inputs(rows)
        b.run(); b.check_r1cs(); b.sync()
        st = b.status()
        if name == "mix":
            assert st[7] & rt.ST_ASSERT_FAILED
            st = np.delete(st, 7)
        assert (st == 0).all(), (name, shape)
        got = b.witnesses()
        for i in (0, 1, 15, 16, 63, 64, 65, 149):
            want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, rows[i]))
            assert failed is None and got[i].tobytes() == b"".join(v.to_bytes(32, "little") for v in want), (name, shape, i)
        b.close(); c.close()


@pytest.mark.gpu
def test_gpu_variant_choice_and_equality_with_the_strand_variants(tmp_path, monkeypatch):
    from circom_amd import runtime as rt
    cp = compile_program(Program(Poseidon(2)), str(tmp_path), "poseidon2", sym=False, pipe=(8, 8))
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    rng = np.random.default_rng(4)
    B = 700
    ins = [[int.from_bytes(rng.bytes(32), "little") % Q for _ in range(2)] for _ in range(B)]
    outs = {}
    for mode in ("default", "0", "1"):
        if mode != "default":
            monkeypatch.setenv("CW_PIPE", mode)
        b = c.batch(B)
        assert (b.pipelined is not None) == (mode != "0")             # a tape that carries the variant uses it for small batches
        b.set_inputs(ins)
        b.run(); b.check_r1cs(); b.sync()
        assert (b.status() == 0).all()
        outs[mode] = b.witnesses().tobytes()
        assert b.signal(699, 1) == poseidon_hash(Q, ins[699])
        b.close()
    assert outs["default"] == outs["0"] == outs["1"]
    c.close()
