"""Bit-plane path (hip_elements/bitblast.py + bitsched.py + csrc/cw_bits.hip): circuits whose signals are all boolean
for 0/1 inputs are evaluated one BIT per signal per instance.  CPU: the gate network and the scheduled program
reproduce the flat witness code (oracle/tape_eval.eval_flat) on random 0/1 inputs; the executor's hazard rules are
enforced by the replay.  GPU: witnesses, status words, R1CS verdicts and every egress path are bit-exact with the
oracle — including instances whose inputs are NOT 0/1 (re-run by the 256-bit schedule)."""
import hashlib
import random

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from circom_amd.circuits.sha256 import Xor3, Maj_t, Ch_t, BinSum, RotR, Sha256
from circom_amd.hip_elements import bitblast as BB, bitmap as BM, bitsched as BS
from oracle.tape_eval import eval_flat, eval_bits, check_r1cs, ScheduleHazard


@template
def BitGadget(c, n):
    """a few SHA-256 building blocks wired together: xor3 of rotations, maj, ch, a 4-operand adder"""
    a = c.input("a", n)
    b = c.input("b", n)
    d = c.input("d", n)
    out = c.output("out", n + 2)
    rot = c.component("rot", RotR(n, 3))
    x3 = c.component("x3", Xor3(n))
    mj = c.component("mj", Maj_t(n))
    ch = c.component("ch", Ch_t(n))
    for k in range(n):
        c.set(rot["in"][k], a[k])
    for k in range(n):
        c.set(x3["a"][k], rot["out"][k]); c.set(x3["b"][k], b[k]); c.set(x3["c"][k], d[k])
        c.set(mj["a"][k], a[k]); c.set(mj["b"][k], b[k]); c.set(mj["c"][k], d[k])
        c.set(ch["a"][k], d[k]); c.set(ch["b"][k], a[k]); c.set(ch["c"][k], b[k])
    s = c.component("sum", BinSum(n, 4))
    for k in range(n):
        c.set(s["in"][0][k], x3["out"][k])
        c.set(s["in"][1][k], mj["out"][k])
        c.set(s["in"][2][k], ch["out"][k])
        c.set(s["in"][3][k], (0xA5A5A5A5 >> k) & 1)
    for k in range(n + 2):
        c.set(out[k], s["out"][k])


@template
def BitAssert(c):
    """`a === b` on two inputs cannot be discharged at compile time: it becomes an assertion gate"""
    a = c.input("a")
    b = c.input("b")
    out = c.output("out")
    c.enforce(a, b)
    c.set(out, a * b)


@template
def BadBit(c, n):
    """witness code (xor) disagrees with the constraint (and): only the R1CS check can notice"""
    a = c.input("a", n)
    b = c.input("b", n)
    out = c.output("out", n)
    for k in range(n):
        c.hint(out[k], a[k] + b[k] - 2 * a[k] * b[k])
        c.enforce(out[k], a[k] * b[k], runtime_check=False)


@template
def BadWeighted(c, n, big):
    """out = a except that bit 2 is flipped when a0 & a1; the (long, linear) constraint sum w_k out_k === sum w_k a_k has
    small weights (3^k: exact 64-bit path of the check) or field-sized ones (2^200 * 3^k: field path)"""
    a = c.input("a", n)
    out = c.output("out", n)
    m = c.signal("m")
    c.set(m, a[0] * a[1])
    lhs = c.const(0)
    rhs = c.const(0)
    for k in range(n):
        if k == 2:
            c.hint(out[k], a[k] + m - 2 * a[k] * m)
        else:
            c.hint(out[k], a[k] + 0)
        w = 3 ** k * ((1 << 200) if big else 1)
        lhs = lhs + out[k] * w
        rhs = rhs + a[k] * w
    c.enforce(lhs, rhs, runtime_check=False)


@template
def BadWords(c, flip, second_only=False):
    """Three 32-bit words x, y, z (bits of consecutive signals) and a copy of each, except that bit `flip` of the copy of y is
    inverted when x0 & x1.  The check rows are the shapes of a BinSum: `sum 2^k x'_k + 2^32 sum 2^k y'_k === sum 2^k x_k +
    2^32 sum 2^k y_k` (whole words in the low and the high half) and `sum 2^k z'_k - sum 2^k y'_k === sum 2^k z_k - sum 2^k y_k`
    (a negative whole word): the word path of the R1CS check (one vector load + bit-matrix transpose per word) must
    notice exactly the instances with the flipped bit, in both halves of the wave."""
    x = c.input("x", 32); y = c.input("y", 32); z = c.input("z", 32)
    xo = c.output("xo", 32); yo = c.output("yo", 32); zo = c.output("zo", 32)
    m = c.signal("m")
    c.set(m, x[0] * x[1])
    for k in range(32):
        c.hint(xo[k], x[k] + 0)
        c.hint(zo[k], z[k] + 0)
        c.hint(yo[k], y[k] + m - 2 * y[k] * m if k == flip else y[k] + 0)
    w = lambda v, sh=0: sum((v[k] * (1 << (k + sh)) for k in range(1, 32)), v[0] * (1 << sh))
    c.enforce(w(xo) + w(y if second_only else yo, 32), w(x) + w(y, 32), runtime_check=False)
    c.enforce(w(zo) - w(yo), w(z) - w(y), runtime_check=False)


def _rand_bits(fc, n, seed):
    r = random.Random(seed)
    return [[r.randrange(2) for _ in range(fc.n_main_inputs)] for _ in range(n)]


def _flat(fc, row):
    sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code,
                            {fc.main_input_start + k: v for k, v in enumerate(row)})
    return sig, failed


def _masks(fc, rows, base=0):
    return {base + fc.main_input_start + k: sum((rows[i][k] & 1) << i for i in range(len(rows))) for k in range(fc.n_main_inputs)}


def _in_masks(fc, rows):
    """bit-table slot of main input k = IN_BASE + k"""
    return {BS.IN_BASE + k: sum((rows[i][k] & 1) << i for i in range(len(rows))) for k in range(fc.n_main_inputs)}


def test_gate_network_reproduces_the_flat_code():
    fc = flatten(Program(BitGadget(16)))
    net = BB.bitblast(fc)
    assert net is not None and net.stats["asserts_left"] == 0 and net.stats["asserts_proved"] > 0
    rows = _rand_bits(fc, 64, 1)
    val = BB.simulate(net, _masks(fc, rows), 64)
    for i in (0, 1, 31, 63):
        sig, failed = _flat(fc, rows[i])
        assert failed is None
        assert [(val[int(net.sig_node[s])] >> i) & 1 for s in range(fc.n_signals)] == sig


def _lower(fc, ring=BS.DEFAULT_RING, cache=BS.DEFAULT_CACHE, **kw):
    net = BB.bitblast(fc)
    pn = BM.map_network(net)
    return net, pn, BS.lower_bits(pn, fc, ring, cache, **kw)


def _replay(bt, fc, rows, width):
    return eval_bits(bt.recs, bt.cmds, bt.ring, bt.cache, bt.n_slots, _in_masks(fc, rows), width)


def test_every_gate_maps_onto_the_two_stage_primitive():
    # all 256 three-input functions have a recipe of at most 3 primitives, and every recipe computes its function
    rec = BM.recipes()
    assert len(rec) == 256
    X, Y, Z = 0xAA, 0xCC, 0xF0

    def ev(tree):
        if tree == 'x':
            return X
        if tree == 'y':
            return Y
        if tree == 'z':
            return Z
        if tree == 0 or tree == 1:
            return 0xFF * tree
        k, p, q, r = tree
        return BM._prim_eval(ev(p), ev(q), ev(r), k)

    for tt, lst in rec.items():
        assert lst and all(ev(e[4]) == tt and e[3] <= 3 for e in lst)
    # a full adder's carry sits ONE level behind its late operand
    assert any(e[2] == 1 for e in rec[0xE8]) and any(e[0] == 1 for e in rec[0xE8])
    fc = flatten(Program(BitGadget(16)))
    net = BB.bitblast(fc)
    pn = BM.map_network(net)
    assert pn.stats["depth"] <= net.stats["depth"] * 1.25 + 2
    rows = _rand_bits(fc, 64, 11)
    v1 = BB.simulate(net, _masks(fc, rows), 64)
    v2 = BM.simulate(pn, _masks(fc, rows), 64)
    assert all(v1[int(net.sig_node[s])] == v2[int(pn.sig_node[s])] for s in range(fc.n_signals))


@pytest.mark.parametrize("ring,cache", [(8, 8), (16, 12), (32, 44), (64, 16)])
def test_scheduled_program_reproduces_the_flat_code(ring, cache):
    fc = flatten(Program(BitGadget(16)))
    net, pn, bt = _lower(fc, ring, cache)
    assert bt.n_slots < fc.n_signals           # copies share the slot of their source
    assert bt.n_vrows % BS.BATCH == 0 and bt.cmds.shape == (bt.n_vrows // BS.BATCH, BS.CMD_WORDS)
    rows = _rand_bits(fc, 64, 2)
    T = _replay(bt, fc, rows, 64)
    for i in (0, 5, 63):
        sig, _ = _flat(fc, rows[i])
        assert [(T[int(bt.sig_slot[s])] >> i) & 1 for s in range(fc.n_signals)] == sig
    # every produced row is flushed exactly once; nothing is flushed onto the constant / input rows
    n_in_rows = (BS.IN_BASE + fc.n_main_inputs + 63) // 64
    flushed = [int(bt.cmds[b, 2 + 2 * BS.MAX_LOADS + 2 * j]) // 512 for b in range(bt.cmds.shape[0])
               for j in range((int(bt.cmds[b, 0]) >> 8) & 0xFF)]
    assert sorted(flushed) == list(range(n_in_rows, bt.n_slots // 64))


def test_whole_words_keep_consecutive_slots():
    # the 32 output bits of a 32-bit adder are consecutive signals: the R1CS check reads them as ONE word
    # (cw_bits_r1cs_int_kernel), so the scheduler keeps their slots together although they are produced in different vrows
    fc = flatten(Program(BitGadget(32)))
    net, pn, bt = _lower(fc)
    assert bt.stats["atoms_kept"] >= 1
    runs = 0
    sl = bt.sig_slot
    for s in range(fc.n_signals - 31):
        if all(int(sl[s + k]) == int(sl[s]) + k for k in range(32)) and int(sl[s]) % 32 == 0 and int(sl[s]) >= 64:
            runs += 1
    assert runs >= 1


def test_replay_rejects_programs_that_break_the_executor_rules():
    fc = flatten(Program(BitGadget(8)))
    net, pn, bt = _lower(fc, 32, 12)
    rows = _rand_bits(fc, 4, 3)
    # an operand that points at a ring entry no vrow has written yet
    recs = bt.recs.copy()
    const_off = (bt.ring + bt.cache) * 512
    idle = const_off | (const_off << 16)
    v = next(v for v in range(bt.n_vrows) if any(int(recs[v * 64 + l, 0]) != idle for l in range(64)))
    assert v + 3 < bt.ring                       # the first operations run as soon as the input rows have arrived
    lane = next(l for l in range(64) if int(recs[v * 64 + l, 0]) != idle)
    recs[v * 64 + lane, 0] = (int(recs[v * 64 + lane, 0]) & 0xFFFF0007) | (((v + 3) % bt.ring) * 512 + 63 * 8)
    with pytest.raises(ScheduleHazard):
        eval_bits(recs, bt.cmds, bt.ring, bt.cache, bt.n_slots, _in_masks(fc, rows), 4)
    # an operand one vrow behind its producer reads the entry BEFORE that write: the value differs from the network's
    recs = bt.recs.copy()
    cand = [(v2, l2) for v2 in range(v + 2, bt.n_vrows) for l2 in range(64) if int(recs[v2 * 64 + l2, 0]) != idle][:200]
    ref = BM.simulate(pn, _masks(fc, rows), 4)
    good = _replay(bt, fc, rows, 4)
    assert all(good[int(bt.sig_slot[s])] == ref[int(pn.sig_node[s])] for s in range(fc.n_signals))
    # a row load of a row that no batch has flushed yet
    cmds = bt.cmds.copy()
    cmds[0, 0] = 1
    cmds[0, 2] = (bt.n_slots // 64 - 1) * 512
    cmds[0, 3] = bt.ring * 512
    with pytest.raises(ScheduleHazard):
        eval_bits(bt.recs, cmds, bt.ring, bt.cache, bt.n_slots, _in_masks(fc, rows), 4)
    # offsets outside the LDS areas, a flush onto the input rows
    recs = bt.recs.copy()
    recs[5, 1] = (int(recs[5, 1]) & 0xFFFF) | ((const_off + 8) << 16)
    with pytest.raises(ScheduleHazard):
        eval_bits(recs, bt.cmds, bt.ring, bt.cache, bt.n_slots, _in_masks(fc, rows), 4)
    cmds = bt.cmds.copy()
    cmds[1, 0] = 1 << 8
    cmds[1, 2 + 2 * BS.MAX_LOADS] = 0
    cmds[1, 3 + 2 * BS.MAX_LOADS] = bt.ring * 512
    with pytest.raises(ScheduleHazard):
        eval_bits(bt.recs, cmds, bt.ring, bt.cache, bt.n_slots, _in_masks(fc, rows), 4)


def test_gate_free_boolean_circuit_gets_no_bit_program():
    # every signal an input bit or a constant: nothing to evaluate (the scheduler divided by zero vrows)
    @template
    def Wires(c, n):
        a = c.input("a", n)
        out = c.output("out", n)
        for k in range(n):
            c.set(out[k], a[k])
    fc = flatten(Program(Wires(80)))
    net = BB.bitblast(fc)
    if net is not None:
        assert BS.lower_bits(BM.map_network(net), fc) is None
    import tempfile
    cp = compile_program(Program(Wires(80)), tempfile.mkdtemp(), "wires", sym=False, strands=(1,), bits=True)
    assert cp.bittape is None


def test_auto_mode_wants_evidence_of_a_bit_level_circuit(tmp_path):
    # Num2Bits on FIELD-valued inputs "is boolean for 0/1 inputs" with almost no gates: auto mode keeps such a
    # circuit on the 256-bit schedule with all its strand variants; SHA-256 (0.7 gates per signal) gets its bit program
    from circom_amd.circuits.basic import Num2Bits

    @template
    def ManyNum2Bits(c, n, w):
        x = c.input("x", n)
        out = c.output("out")
        cs = [c.component("n2b%d" % i, Num2Bits(w)) for i in range(n)]
        for i in range(n):
            c.set(cs[i]["in"], x[i])
        c.set(out, cs[0]["out"][0] * cs[1]["out"][0])

    prog = Program(ManyNum2Bits(80, 64))
    fc = flatten(prog)
    assert fc.n_signals > 4096
    cp = compile_program(prog, str(tmp_path / "a"), "n2b", sym=False, strands=(1, 4))
    assert cp.bittape is None
    from circom_amd import runtime as rt
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert c.bits_info() == {}
    c.close()
    import os
    os.environ["CW_BITS"] = "0"                      # compile-time switch: no bit program in the tape at all
    try:
        cp2 = compile_program(Program(BitGadget(8)), str(tmp_path / "b"), "bg", sym=False, strands=(1,), bits=True)
    finally:
        del os.environ["CW_BITS"]
    assert cp2.bittape is None


def test_unprovable_assert_becomes_an_assertion_slot():
    fc = flatten(Program(BitAssert()))
    net = BB.bitblast(fc)
    assert net is not None and len(net.asserts) == 1
    net, pn, bt = _lower(fc)
    assert len(bt.assert_slots) == 1
    rows = [[0, 0], [0, 1], [1, 0], [1, 1]]
    T = _replay(bt, fc, rows, 4)
    viol = T[int(bt.assert_slots[0])]
    assert viol == 0b0110
    for i, row in enumerate(rows):
        sig, failed = _flat(fc, row)
        assert (failed is not None) == bool((viol >> i) & 1)


def test_arithmetic_circuits_are_left_to_the_wide_schedule():
    from circom_amd.circuits.poseidon import Poseidon
    from circom_amd.circuits.basic import Num2Bits, IsZero
    assert BB.bitblast(flatten(Program(Poseidon(2)))) is None
    assert BB.bitblast(flatten(Program(IsZero()))) is None            # division / select
    # small circuits never get a bit program unless asked for
    import tempfile
    cp = compile_program(Program(BitGadget(8)), tempfile.mkdtemp(), "bg", sym=False, strands=(1,))
    assert cp.bittape is None
    cp = compile_program(Program(BitGadget(8)), tempfile.mkdtemp(), "bg", sym=False, strands=(1,), bits=True)
    assert cp.bittape is not None


def test_sha256_one_block_bitplane_digest():
    fc = flatten(Program(Sha256(64)))
    net, pn, bt = _lower(fc)
    assert net.stats["asserts_left"] == 0
    msgs = [b"abcdefgh", b"\x00" * 8, b"\xff" * 8, b"MI355X!!"]
    rows = [[(m[i // 8] >> (7 - i % 8)) & 1 for i in range(64)] for m in msgs]
    T = _replay(bt, fc, rows, len(msgs))
    for i, m in enumerate(msgs):
        dg = hashlib.sha256(m).digest()
        want = [(dg[k // 8] >> (7 - k % 8)) & 1 for k in range(256)]
        assert [(T[int(bt.sig_slot[1 + k])] >> i) & 1 for k in range(256)] == want
    sig, _ = _flat(fc, rows[3])
    assert [(T[int(bt.sig_slot[s])] >> 3) & 1 for s in range(fc.n_signals)] == sig
    # vector memory moves whole rows only: a handful of row loads (the main inputs), one flush per produced row
    assert bt.stats["row_loads"] < 0.1 * bt.stats["row_flushes"]


def test_loader_validates_the_bit_program(tmp_path):
    from circom_amd import runtime as rt
    cp = compile_program(Program(BitGadget(8)), str(tmp_path), "bg", sym=False, strands=(1,), bits=True)
    rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path).close()
    tape = bytearray(open(cp.tape_path, "rb").read())
    bt = cp.bittape
    n_cmd = bt.cmds.size * 4
    n_rec = bt.n_vrows * 64 * 8
    tail = 4 * cp.flat.n_signals + 4 * len(bt.assert_slots)
    rec0 = len(tape) - tail - n_cmd - n_rec
    cmd0 = len(tape) - tail - n_cmd
    const_off = (bt.ring + bt.cache) * 512

    def expect_rejected(mutated):
        p = tmp_path / "bad.cwt"
        p.write_bytes(bytes(mutated))
        with pytest.raises(rt.CwError):
            rt.Circuit(p, cp.dat_path, cp.r1cs_path)

    for word, val in ((0, const_off + 16),                                       # operand a beyond the LDS areas
                      (0, (const_off + 16) << 16),                               # operand b beyond the LDS areas
                      (0, 4),                                                    # reserved record bit
                      (1, const_off + 16),                                       # operand c beyond the LDS areas
                      (1, const_off << 16),                                      # result onto the constants
                      (1, 4 << 16)):                                             # misaligned result
        bad = bytearray(tape)
        at = rec0 + 70 * 8 + word * 4
        bad[at:at + 4] = int(val).to_bytes(4, "little")
        expect_rejected(bad)
    for words in ({0: 5},                                                        # too many loads
                  {0: 7 << 8},                                                   # too many flushes
                  {0: 1, 2: bt.n_slots * 8, 3: bt.ring * 512},                   # load of a row beyond the table
                  {0: 1, 2: 0, 3: 0},                                            # load into the ring area
                  {0: 1 << 8, 2 + 2 * BS.MAX_LOADS: 0, 3 + 2 * BS.MAX_LOADS: bt.ring * 512},   # flush onto the input rows
                  {0: 1 << 8, 2 + 2 * BS.MAX_LOADS: (bt.n_slots // 64 - 1) * 512, 3 + 2 * BS.MAX_LOADS: const_off}):   # cache slot beyond the cache
        bad = bytearray(tape)
        for w, val in words.items():
            bad[cmd0 + w * 4: cmd0 + w * 4 + 4] = int(val).to_bytes(4, "little")
        expect_rejected(bad)
    bad = bytearray(tape)                                                        # a signal mapped beyond the table
    at = len(tape) - tail
    bad[at:at + 4] = int(bt.n_slots).to_bytes(4, "little")
    expect_rejected(bad)


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(params=["interp", "jit"])
def engine(request, monkeypatch):
    """every GPU test of this file runs on both engines of the bit-plane path: the interpreter (cw_bits_eval_kernel) and the
    circuit's emitted gfx950 code (hip_elements/bitjit.py; CW_BITS_JIT=1 selects it below its batch threshold)"""
    monkeypatch.setenv("CW_BITS_JIT", "1" if request.param == "jit" else "0")
    return request.param


def _gpu(tmp_path, prog, name, **kw):
    from circom_amd import runtime as rt
    cp = compile_program(prog, str(tmp_path), name, sym=False, strands=(1,), bits=True, jit=True, **kw)
    assert cp.bittape is not None and cp.jit is not None
    return cp, rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)


@pytest.mark.gpu
def test_gpu_bitplane_matches_oracle_on_every_instance(tmp_path, engine):
    cp, c = _gpu(tmp_path, Program(BitGadget(16)), "bg16")
    fc = cp.flat
    B = 300                                                    # ragged last group
    rows = _rand_bits(fc, B, 7)
    b = c.batch(B)
    assert b.bitmode and b.jit == (engine == "jit")
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    bulk = b.witnesses()
    pub = b.public_signals()
    for i in range(B):
        sig, failed = _flat(fc, rows[i])
        assert failed is None
        got = [int.from_bytes(bulk[i, k].tobytes(), "little") for k in range(fc.n_signals)]
        assert got == sig, i
        assert [int.from_bytes(pub[i, k].tobytes(), "little") for k in range(c.n_public)] == sig[1:1 + c.n_public]
    assert b.witness(299) == _flat(fc, rows[299])[0]
    assert b.signal(17, 5) == _flat(fc, rows[17])[0][5]
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_ingest_and_egress_orders_at_many_groups(tmp_path, engine):
    """round 5's memory orders: the 32-byte ingest walks full groups from a per-group rotated first instance (four waves on
    consecutive 256-input chunks; 384 inputs = a full and a half chunk; the ragged last group takes the plain loop), the egress
    walks the GROUPS fastest from 8 groups per launch on (18 here; 3 205 witness elements = three full element blocks and a
    ragged one).  Every instance against the oracle, unaligned windows against the bulk image, the per-instance path (one
    group per launch: elements fastest) against both."""
    cp, c = _gpu(tmp_path, Program(BitGadget(128)), "bg128")
    fc = cp.flat
    B = 1100
    rows = _rand_bits(fc, B, 21)
    rows[700][5] = 2                                           # not a bit, in a full group: flagged by its own lane, re-run wide
    rows[1099][383] = c.q - 1                                  # and in the ragged group, last input
    b = c.batch(B)
    assert b.bitmode and b.jit == (engine == "jit")
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    bulk = b.witnesses()
    assert bulk.shape == (B, fc.n_signals, 32)
    vals = bulk[:, :, :8].copy().view(np.uint64)[:, :, 0]
    for i in range(B):
        sig, failed = _flat(fc, rows[i])
        if failed is not None:
            assert st[i] & 1, i
            continue
        assert not (st[i] & 1), i
        if i in (700, 1099):
            assert [int.from_bytes(bulk[i, k].tobytes(), "little") for k in range(fc.n_signals)] == sig, i
        else:
            assert not bulk[i, :, 8:].any() and vals[i].tolist() == sig, i
    for first, n in ((37, 900), (64, 576), (1, 1099), (1000, 100)):
        assert (b.witnesses(first, n) == bulk[first:first + n]).all(), (first, n)
    for i in (0, 63, 64, 699, 701, 1098):
        assert b.witness(i) == vals[i].tolist()
    pub = b.public_signals()
    assert (pub == bulk[:, 1:1 + c.n_public]).all()
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_non_boolean_inputs_are_rerun_by_the_wide_schedule(tmp_path, engine):
    cp, c = _gpu(tmp_path, Program(BitGadget(8)), "bg8")
    fc = cp.flat
    q = c.q
    B = 130
    rows = _rand_bits(fc, B, 9)
    odd = {3: (0, 2), 64: (5, q - 1), 65: (23, 12345678901234567890), 129: (7, 1 << 200)}
    for i, (k, v) in odd.items():
        rows[i][k] = v
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    bulk = b.witnesses()
    pub = b.public_signals()
    for i in range(B):
        sig, failed = _flat(fc, rows[i])
        if failed is not None:
            assert st[i] & 1, i                                 # the reference would abort on this input (assert)
            continue
        assert not (st[i] & 1), i
        assert [int.from_bytes(bulk[i, k].tobytes(), "little") for k in range(fc.n_signals)] == sig, i
        assert b.witness(i) == sig
        assert [int.from_bytes(pub[i, k].tobytes(), "little") for k in range(c.n_public)] == sig[1:1 + c.n_public]
        assert bool(st[i] & 4) == (check_r1cs(q, fc.constraints, sig) is not None)
    # a second run with clean inputs drops the side batch
    rows2 = _rand_bits(fc, B, 10)
    b.set_inputs(rows2)
    b.run(); b.sync()
    assert (b.status() == 0).all()
    assert b.witness(64) == _flat(fc, rows2[64])[0]
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_assertion_gate_flags_and_reruns(tmp_path, engine):
    cp, c = _gpu(tmp_path, Program(BitAssert()), "bassert")
    rows = [[i & 1, (i >> 1) & 1] for i in range(200)]
    b = c.batch(200)
    b.set_inputs(rows)
    b.run(); b.sync()
    st = b.status()
    for i, (x, y) in enumerate(rows):
        assert bool(st[i] & 1) == (x != y), i
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_bitplane_r1cs_check_reports_first_bad_row(tmp_path, engine):
    cp, c = _gpu(tmp_path, Program(BadBit(12)), "badbit")
    fc = cp.flat
    B = 200
    rows = _rand_bits(fc, B, 11)
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st, fb = b.status(), b.r1cs_first_bad()
    for i in range(B):
        w = b.witness(i)
        want = check_r1cs(c.q, fc.constraints, w)
        assert bool(st[i] & 4) == (want is not None), i
        if want is not None:
            assert fb[i] == want, (i, fb[i], want)
    b.close(); c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("big", [False, True])
def test_gpu_bitplane_r1cs_long_rows(tmp_path, engine, big):
    cp, c = _gpu(tmp_path, Program(BadWeighted(12, big)), "badw%d" % big)
    fc = cp.flat
    B = 200
    rows = _rand_bits(fc, B, 12)
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st, fb = b.status(), b.r1cs_first_bad()
    n_bad = 0
    for i in range(B):
        w = b.witness(i)
        assert w == _flat(fc, rows[i])[0]
        want = check_r1cs(c.q, fc.constraints, w)
        assert (want is not None) == bool(rows[i][0] & rows[i][1])
        assert bool(st[i] & 4) == (want is not None), i
        if want is not None:
            assert fb[i] == want, (i, fb[i], want)
            n_bad += 1
    assert 0 < n_bad < B
    b.close(); c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("flip,second_only", [(0, False), (13, True), (31, False), (31, True)])
def test_gpu_bitplane_r1cs_whole_words(tmp_path, engine, flip, second_only):
    cp, c = _gpu(tmp_path, Program(BadWords(flip, second_only)), "badwords%d" % flip)
    fc = cp.flat
    B = 200
    rows = _rand_bits(fc, B, 40 + flip)
    b = c.batch(B)
    assert b.bitmode and b.jit == (engine == "jit")
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st, fb = b.status(), b.r1cs_first_bad()
    n_bad = 0
    for i in range(B):
        w = b.witness(i)
        assert w == _flat(fc, rows[i])[0]
        want = check_r1cs(c.q, fc.constraints, w)
        assert (want is not None) == bool(rows[i][0] & rows[i][1])
        assert bool(st[i] & 4) == (want is not None), i
        if want is not None:
            assert fb[i] == want == (2 if second_only else 1), (i, fb[i], want)
            n_bad += 1
    assert 0 < n_bad < B
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_emitted_audit_of_a_table_somebody_changed(tmp_path, monkeypatch):
    """the stand-alone audit of an emitted-code batch is emitted code too (bitjit.lower_jit(audit_of=): the check's gates on
    rows LOADED from the table): clean on the table the evaluation left, and after a caller took the raw table (cw_device_bits)
    and flipped one wire in three instances, exactly the instances whose witness now violates a constraint are flagged, with
    the first violated row - by the emitted audit and by the general kernels (CW_R1CS_AUDIT_GENERAL=1) alike"""
    import ctypes as C
    monkeypatch.setenv("CW_BITS_JIT", "1")
    cp, c = _gpu(tmp_path, Program(BitGadget(16)), "bg16a")
    assert cp.jit.audit_code and cp.jit.check_complete
    fc = cp.flat
    B = 300
    rows = _rand_bits(fc, B, 21)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    results = []
    for general in (False, True):
        if general:
            monkeypatch.setenv("CW_R1CS_AUDIT_GENERAL", "1")
        b = c.batch(B)
        assert b.jit
        b.set_inputs(rows)
        b.run(); b.check_r1cs(); b.sync()
        assert (b.status() == 0).all()
        ptr, nbytes, spg = b.device_bits()                         # from here on every check audits the table
        assert ptr and spg == b.bits_slots
        b.check_r1cs(); b.sync()
        assert (b.status() == 0).all()                              # the untouched table passes its audit
        # flip the value of one constrained wire in instances 5, 70 and 299
        victim = next(s_ for s_ in range(fc.n_signals - 1, 0, -1) if any(s_ in A or s_ in B_ or s_ in C_ for A, B_, C_ in fc.constraints)
                      and not fc.main_input_start <= s_ < fc.main_input_start + fc.n_main_inputs)
        slot = int(b.signal_slots()[victim])
        for i in (5, 70, 299):
            off = 8 * b.bits_index(i // 64, slot)
            word = C.c_uint64()
            assert hip.hipMemcpy(C.byref(word), C.c_void_p(ptr + off), 8, 2) == 0
            word.value ^= 1 << (i % 64)
            assert hip.hipMemcpy(C.c_void_p(ptr + off), C.byref(word), 8, 1) == 0
        b.check_r1cs(); b.sync()
        st, fb = b.status(), b.r1cs_first_bad()
        n_bad = 0
        for i in range(B):
            w = b.witness(i)
            want = check_r1cs(c.q, fc.constraints, w)
            assert bool(st[i] & 4) == (want is not None), (general, i)
            if want is not None:
                assert fb[i] == want and i in (5, 70, 299)
                n_bad += 1
        assert n_bad >= 1
        results.append((st.tolist(), fb.tolist()))
        b.close()
    assert results[0] == results[1]
    c.close()


@pytest.mark.gpu
def test_gpu_emitted_code_beyond_one_chunk_when_the_audit_spills(tmp_path, monkeypatch):
    """the audit program of Sha256(64) spills scratch rows behind the table, which raises the rows per chunk -
    the stride every kernel walks the table with, and an immediate of BOTH emitted code objects.  Three chunks (the second and
    third sit at base + k * stride): every digest against hashlib, full witnesses from every chunk against the oracle, and
    the emitted audit of the table (CW_R1CS_AUDIT=1) clean."""
    import hashlib
    monkeypatch.setenv("CW_BITS_JIT", "1")
    monkeypatch.setenv("CW_JIT_AUDIT_REGS", "64,8")          # a starved audit: its spills add scratch rows behind the table
    cp, c = _gpu(tmp_path, Program(Sha256(64)), "sha256_64s")
    assert cp.jit.audit_code and cp.jit.n_slots > cp.jit.stats["slots"], "this circuit's audit was expected to add scratch rows"
    assert cp.jit.code_stride == cp.jit.audit_stride == cp.jit.n_slots * 256
    fc = cp.flat
    B = 2048 * 2 + 700
    rng = np.random.default_rng(3)
    msgs = rng.integers(0, 256, size=(B, 8), dtype=np.uint8)
    rows = np.unpackbits(msgs, axis=1)                               # msb first, as the circuit's inputs
    b = c.batch(B)
    assert b.jit and b.bits_slots == cp.jit.n_slots
    b.set_inputs(rows.tolist())
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    pub = b.public_signals()
    assert not pub[:, :256, 1:].any()
    got = np.packbits(pub[:, :256, 0], axis=1)
    want = np.frombuffer(b"".join(hashlib.sha256(m.tobytes()).digest() for m in msgs), dtype=np.uint8).reshape(B, 32)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "digest of instance %d (chunk %d) differs from hashlib" % (int(bad[0]), int(bad[0]) // 2048)
    for i in (0, 2047, 2048, 4095, 4096, B - 1):
        sig, failed = _flat(fc, rows[i].tolist())
        assert failed is None and b.witness(i) == sig, i
    monkeypatch.setenv("CW_R1CS_AUDIT", "1")
    b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_looped_body_of_a_repeated_template(tmp_path, monkeypatch):
    """ONE body per repeated template on the device (bitjit.lower_jit(loop=True); the reference's `<T>_<id>_run`,
    template.rs:160-474): the three compression blocks of Sha256(1024) run as three iterations of one 346 K-instruction body -
    rows relative to the iteration's base (s40), the block's inputs (IV constants / previous digest, message bits / padding)
    through the iteration's table of row offsets behind the code.  Three chunks of instances: every digest against hashlib,
    full witnesses of every chunk against the oracle, the fused check clean, the LOOPED audit of the table clean - and after a
    caller flipped one wire of the last block in three instances, the audit flags exactly the instances that now violate."""
    import ctypes as C
    import hashlib
    monkeypatch.setenv("CW_BITS_JIT", "1")
    monkeypatch.setenv("CW_JIT_LOOP", "1")
    cp, c = _gpu(tmp_path, Program(Sha256(1024)), "sha256_1024l")
    assert cp.jit.loop and cp.jit.loop["K"] == 3 and cp.jit.stats["loop"]["external_values"] == 768
    assert cp.jit.audit_code and len(cp.jit.code) < 4 << 20 and cp.jit.check_complete
    fc = cp.flat
    B = 2048 * 2 + 300
    rng = np.random.default_rng(11)
    msgs = rng.integers(0, 256, size=(B, 128), dtype=np.uint8)
    rows = np.unpackbits(msgs, axis=1)
    b = c.batch(B)
    assert b.jit
    b.set_inputs(rows.tolist())
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    pub = b.public_signals()
    assert not pub[:, :256, 1:].any()
    got = np.packbits(pub[:, :256, 0], axis=1)
    want = np.frombuffer(b"".join(hashlib.sha256(m.tobytes()).digest() for m in msgs), dtype=np.uint8).reshape(B, 32)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "digest of instance %d (chunk %d) differs from hashlib" % (int(bad[0]), int(bad[0]) // 2048)
    for i in (0, 2047, 2048, 4096, B - 1):
        sig, failed = _flat(fc, rows[i].tolist())
        assert failed is None and b.witness(i) == sig, i
    monkeypatch.setenv("CW_R1CS_AUDIT", "1")
    b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    # a wire of the LAST block (third iteration), flipped in one instance of every chunk
    ptr, nbytes, spg = b.device_bits()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    victim = next(s_ for s_ in range(fc.n_signals - 2000, 0, -1) if any(s_ in A or s_ in B_ or s_ in C_ for A, B_, C_ in fc.constraints[-20000:]))
    slot = int(b.signal_slots()[victim])
    assert slot >= cp.jit.loop["base"] + 2 * cp.jit.loop["R"], "the victim was meant to be a row of the third iteration"
    for i in (7, 2048 + 70, B - 2):
        off = 8 * b.bits_index(i // 64, slot)
        word = C.c_uint64()
        assert hip.hipMemcpy(C.byref(word), C.c_void_p(ptr + off), 8, 2) == 0
        word.value ^= 1 << (i % 64)
        assert hip.hipMemcpy(C.c_void_p(ptr + off), C.byref(word), 8, 1) == 0
    b.check_r1cs(); b.sync()
    st, fb = b.status(), b.r1cs_first_bad()
    flagged = np.nonzero(st & 4)[0].tolist()
    assert flagged == [7, 2048 + 70, B - 2], flagged
    for i in flagged:
        assert fb[i] == check_r1cs(c.q, fc.constraints, b.witness(i))
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_sha256_two_blocks_bitplane(tmp_path, engine):
    cp, c = _gpu(tmp_path, Program(Sha256(512)), "sha256_512")
    fc = cp.flat
    B = 200
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2, size=(B, 512), dtype=np.uint8)
    arr = np.zeros((B, 512, 32), dtype=np.uint8)
    arr[:, :, 0] = bits
    b = c.batch(B)
    assert b.bitmode and b.jit == (engine == "jit")
    b.set_inputs(arr)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    pub = b.public_signals()
    for i in range(B):
        dg = np.unpackbits(np.frombuffer(hashlib.sha256(np.packbits(bits[i]).tobytes()).digest(), dtype=np.uint8))
        assert (pub[i, :256, 0] == dg).all() and not pub[i, :256, 1:].any(), i
    for i in (0, 199):
        sig, failed = _flat(fc, bits[i].tolist())
        assert failed is None and b.witness(i) == sig
    b.close(); c.close()


class _Hip:
    """device buffers for the tests without a second HIP runtime in the process (torch bundles its own)"""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.h = C.CDLL("libamdhip64.so")
        self.h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.h.hipFree.argtypes = [C.c_void_p]

    def alloc(self, n):
        p = self.C.c_void_p()
        assert self.h.hipMalloc(self.C.byref(p), n) == 0
        return p.value

    def upload(self, arr):
        p = self.alloc(arr.nbytes)
        assert self.h.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0
        return p

    def download(self, p, shape, dtype=np.uint8):
        out = np.zeros(shape, dtype=dtype)
        assert self.h.hipDeviceSynchronize() == 0
        assert self.h.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0
        return out


@pytest.mark.gpu
def test_gpu_packed_inputs_and_chunked_egress(tmp_path, engine):
    """cw_set_inputs_bits(_device): one bit per input and instance in, the same witnesses out; cw_stream_witnesses_device:
    the chunks, concatenated, are the image cw_get_witnesses returns (odd first/count/chunk: windows straddle groups)."""
    cp, c = _gpu(tmp_path, Program(BitGadget(16)), "bg16p")
    hip = _Hip()
    fc = cp.flat
    B = 200
    rows = _rand_bits(fc, B, 21)
    masks = np.zeros(((B + 63) // 64, fc.n_main_inputs), dtype=np.uint64)
    for i, row in enumerate(rows):
        for k, v in enumerate(row):
            if v:
                masks[i // 64, k] |= np.uint64(1) << np.uint64(i % 64)
    masks[-1] |= np.uint64(0xFF) << np.uint64(56)                 # garbage beyond the batch in the last group: ignored
    b = c.batch(B)
    b.set_inputs_bits(masks)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    bulk = b.witnesses()
    for i in (0, 63, 64, 199):
        sig, _ = _flat(fc, rows[i])
        assert [int.from_bytes(bulk[i, k].tobytes(), "little") for k in range(fc.n_signals)] == sig, i
    # the same from a device buffer
    b2 = c.batch(B)
    b2.set_inputs_bits_device(hip.upload(masks))
    b2.run(); b2.sync()
    assert (b2.witnesses() == bulk).all()
    # chunked egress
    first, count, chunk = 37, 150, 41
    nbytes = chunk * c.n_witness * 32
    bufs = [hip.alloc(nbytes), hip.alloc(nbytes)]
    got = np.zeros((count, c.n_witness, 32), dtype=np.uint8)
    seen = []

    def consume(f, n, ptr, stream):
        assert ptr in bufs                                     # a test consumer: simply wait and copy out
        got[f - first:f - first + n] = hip.download(ptr, (chunk, c.n_witness, 32))[:n]
        seen.append((f, n, bufs.index(ptr)))
        return 0

    b.stream_witnesses_device(first, count, chunk, bufs[0], bufs[1], consume)
    assert seen == [(37, 41, 0), (78, 41, 1), (119, 41, 0), (160, 27, 1)]
    assert (got == bulk[first:first + count]).all()
    # a 256-bit batch refuses packed inputs
    import os
    os.environ["CW_BITS"] = "0"
    try:
        b3 = c.batch(64)
    finally:
        del os.environ["CW_BITS"]
    assert not b3.bitmode
    with pytest.raises(Exception):
        b3.set_inputs_bits(masks[:1])
    b.close(); b2.close(); b3.close(); c.close()


@pytest.mark.gpu
def test_gpu_compact_container_round_trips_to_the_wtns_files(tmp_path, engine):
    """cw_write_wtnsb: one file for the batch (bit planes + slot map + the field elements of the instances the 256-bit schedule
    re-ran); circom_amd/wtnsb.py expands it to exactly the bytes cw_write_wtns writes, for boolean and non-boolean instances"""
    from circom_amd import wtnsb
    cp, c = _gpu(tmp_path, Program(BitGadget(16)), "bg16w")
    fc = cp.flat
    B = 150
    rows = _rand_bits(fc, B, 31)
    rows[70][3] = 5                                              # not a bit: re-run wide, carried as field elements
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    p = tmp_path / "batch.wtnsb"
    b.write_wtnsb(p)
    w = wtnsb.load(p)
    assert (w.kind, w.batch, w.n_witness, w.prime) == (1, B, c.n_witness, c.q) and 70 in w.wide
    assert w.shift == (5 if engine == "jit" else 0)
    for i in (0, 63, 64, 70, 149):
        q = tmp_path / ("i%d.wtns" % i)
        b.write_wtns(i, q)
        assert w.expand(i) == q.read_bytes(), i
    # the container is the compact form: ~1 bit per signal value and instance against 32 bytes per element
    assert p.stat().st_size < B * c.n_witness * 32 // 4
    b.close(); c.close()
