"""Front-end + writers against the only golden vectors the reference holds for this path:
the docs' basic.circom R1CS (--O0), its .sym, Multiplier2 3*11=33, plus format round trips."""
import json
import struct

import numpy as np

from circom_amd.frontend.dsl import Program
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements.lower import lower
from circom_amd.hip_elements import writers
from circom_amd.circuits.basic import BasicMain, Multiplier2, Num2Bits, IsZero
from circom_amd.field import fp_for
from oracle.field import Field, PRIMES
from oracle.tape_eval import eval_flat, eval_rows, eval_tape, check_r1cs

Q = PRIMES["bn128"]
QM1 = str(Q - 1)

# mkdocs/docs/circom-language/formats/constraints-json.md:84-93 (--O0 listing)
GOLDEN_O0 = [
    [{}, {}, {"2": "1", "5": QM1}],
    [{}, {}, {"0": "1", "2": "2", "3": "1", "6": QM1}],
    [{}, {}, {"1": QM1, "4": "1"}],
    [{"5": QM1}, {"6": "1"}, {"4": QM1}],
]
# mkdocs/docs/circom-language/formats/sym.md:67-74 (--O0)
GOLDEN_SYM_O0 = "1,1,1,main.out\n2,2,1,main.in[0]\n3,3,1,main.in[1]\n4,4,0,main.c.out\n5,5,0,main.c.in[0]\n6,6,0,main.c.in[1]\n"


def test_basic_circom_constraints_match_docs_golden():
    fc = flatten(Program(BasicMain()))
    got = [[{str(k): str(v) for k, v in part.items()} for part in con] for con in fc.constraints]
    assert got == GOLDEN_O0


def test_basic_circom_sym_matches_docs_golden(tmp_path):
    fc = flatten(Program(BasicMain()))
    writers.write_sym(tmp_path / "b.sym", fc)
    assert (tmp_path / "b.sym").read_text() == GOLDEN_SYM_O0


def test_multiplier2_numbering_dat_and_wtns(tmp_path):
    fc = flatten(Program(Multiplier2()))
    # 4 signals, inputs start at 2, witness = 1, c, a, b
    assert fc.n_signals == 4 and fc.main_input_start == 2 and fc.inputs == [("a", 2, 1), ("b", 3, 1)]
    sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {2: 3, 3: 11})
    assert failed is None and sig == [1, 33, 3, 11]
    size = writers.write_dat(tmp_path / "m.dat", fc)
    raw = (tmp_path / "m.dat").read_bytes()
    assert size == 256 and len(raw) == 256 * 24 + 4 * 8      # no constants
    ha, hb = writers.fnv1a("a"), writers.fnv1a("b")
    assert struct.unpack_from("<QQQ", raw, (ha % 256) * 24) == (ha, 2, 1)
    assert struct.unpack_from("<QQQ", raw, (hb % 256) * 24) == (hb, 3, 1)
    assert list(np.frombuffer(raw[256 * 24:], dtype="<u8")) == [0, 1, 2, 3]
    w = writers.wtns_bytes(Q, sig)
    assert len(w) == 204 and w[:4] == b"wtns" and w[28:60] == Q.to_bytes(32, "little")
    assert w[76 + 32:76 + 64] == (33).to_bytes(32, "little")


def test_fnv1a_known_values():
    assert writers.fnv1a("") == 0xCBF29CE484222325
    assert writers.fnv1a("a") == 0xAF63DC4C8601EC8C          # published FNV-1a 64 test vector


def test_dat_constant_encoding():
    fp = fp_for("bn128")
    # short constant: int32 value, tag 0x40000000, Montgomery limbs (c_code_generator.rs:640-677)
    b = writers.dat_constant(5, fp)
    assert len(b) == 40 and struct.unpack_from("<iI", b) == (5, 0x40000000)
    assert int.from_bytes(b[8:], "little") == 5 * (1 << 256) % Q
    b = writers.dat_constant(Q - 3, fp)
    assert struct.unpack_from("<iI", b) == (-3, 0x40000000)
    b = writers.dat_constant(1 << 100, fp)
    assert struct.unpack_from("<iI", b) == (0, 0xC0000000)
    assert int.from_bytes(b[8:], "little") == (1 << 356) % Q


def test_r1cs_file_layout(tmp_path):
    fc = flatten(Program(BasicMain()))
    writers.write_r1cs(tmp_path / "b.r1cs", fc)
    raw = (tmp_path / "b.r1cs").read_bytes()
    assert raw[:4] == b"r1cs" and struct.unpack_from("<II", raw, 4) == (1, 3)
    # section order on disk is 2, 1, 3 (dag/src/r1cs_porting.rs:13-46)
    off = 12
    order = []
    secs = {}
    while off < len(raw):
        typ, ln = struct.unpack_from("<IQ", raw, off)
        order.append(typ)
        secs[typ] = raw[off + 12:off + 12 + ln]
        off += 12 + ln
    assert order == [2, 1, 3]
    h = secs[1]
    assert struct.unpack_from("<I", h)[0] == 32 and int.from_bytes(h[4:36], "little") == Q
    assert struct.unpack_from("<IIIIQI", h, 36) == (7, 1, 0, 2, 7, 4)
    assert list(np.frombuffer(secs[3], dtype="<u8")) == list(range(7))
    # first constraint: A = {}, B = {}, C = {2: 1, 5: q-1}
    c = secs[2]
    assert struct.unpack_from("<II", c) == (0, 0)
    assert struct.unpack_from("<I", c, 8)[0] == 2
    assert struct.unpack_from("<I", c, 12)[0] == 2 and int.from_bytes(c[16:48], "little") == 1
    assert struct.unpack_from("<I", c, 48)[0] == 5 and int.from_bytes(c[52:84], "little") == Q - 1


def test_lowered_schedule_equals_flat_semantics():
    import random
    rng = random.Random(3)
    for prog, slots, small in ((Program(BasicMain()), [2, 3], False), (Program(Num2Bits(16)), [17], True),
                               (Program(IsZero()), [2], False)):
        fc = flatten(prog)
        t = lower(fc)
        for trial in range(6):
            inp = {s: (rng.randrange(1 << 16) if small else rng.randrange(Q)) for s in slots}
            if trial == 0:
                inp = {s: 0 for s in slots}
            a, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
            b, st = eval_tape(t, inp)
            assert a == b and (st == 0) == (failed is None)
            if failed is None:
                assert check_r1cs(Q, fc.constraints, a) is None


def test_host_field_matches_oracle_field():
    import random
    rng = random.Random(11)
    fo, fh = Field(Q), fp_for("bn128")
    vals = [0, 1, 2, Q - 1, Q // 2, Q // 2 + 1, 253, 254, 255, Q - 254, Q - 253] + [rng.randrange(Q) for _ in range(60)]
    for name in ("add", "sub", "mul", "div", "pow", "shl", "shr", "band", "bor", "bxor", "eq", "neq", "lt", "gt", "leq",
                 "geq", "land", "lor"):
        for a in vals:
            for b in vals[:20]:
                assert getattr(fo, name)(a, b) == getattr(fh, name)(a, b), (name, a, b)
    for a in vals:
        assert fo.neg(a) == fh.neg(a) and fo.bnot(a) == fh.bnot(a) and fo.inv(a) == fh.inv(a)


def test_strand_schedules_are_race_free_and_equivalent():
    """Multi-strand lowering (waves of one workgroup sharing 64 instances): the prefetch-exact simulator
    raises ScheduleHazard on any cross-strand race inside a barrier epoch or stale one-row-ahead prefetch."""
    import random
    from circom_amd.circuits.poseidon import Poseidon
    rng = random.Random(9)
    for prog, slots, small in ((Program(Poseidon(2)), [2, 3], False), (Program(Num2Bits(16)), [17], True),
                               (Program(IsZero()), [2], False), (Program(BasicMain()), [2, 3], False)):
        fc = flatten(prog)
        for S in (1, 2, 4, 16):
            t = lower(fc, n_strands=S)
            assert len(t.stream_off) == S + 1 and t.stream_off[-1] == len(t.rows)
            for trial in range(2):
                inp = {s: (rng.randrange(1 << 16) if small else rng.randrange(Q)) for s in slots}
                a, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
                b, st = eval_tape(t, inp)
                assert a == b and (st == 0) == (failed is None), (prog.main.name, S)


def test_value_wired_into_thousands_of_places_overflows_the_extra_field_gracefully():
    """One value copied 9000 times (a selector bit fed to every element of a wide mux): the copies are folded into
    the producing row up to the capacity of its extra-destination field, the rest chain through copy rows."""
    from circom_amd.frontend.dsl import template as _t

    @_t
    def Wide(c, n):
        a = c.input("a"); b = c.input("b")
        out = c.output("out", n)
        x = c.signal("x")
        c.set(x, a * b)
        for i in range(n):
            c.set(out[i], x)

    fc = flatten(Program(Wide(9000)))
    for S in (1, 4):
        t = lower(fc, n_strands=S)
        assert t.stats["copies_elided"] >= 8990 and t.stats["copy"] >= 2
        got, st = eval_tape(t, {fc.main_input_start: 6, fc.main_input_start + 1: 7})
        assert st == 0 and got[1:9001] == [42] * 9000
