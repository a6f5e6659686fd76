"""Emitted gate code (hip_elements/bitjit.py): the bit-plane program as straight-line gfx950 code, one `v_bitop3_b32` per
gate, with the R1CS check fused in.  CPU: the IR is executed by oracle/jit_eval.py (poisoned registers, in-order memory
queue) and must reproduce the flat witness code for every signal under generous AND starved register files; the fused check
must flag exactly the instances whose witness violates a constraint; the assembler accepts the text and the C-ABI loader
the section.  The GPU side of the same engine runs through tests/test_bitplane.py (every GPU test there is parametrised over
both engines) and tests/test_baseline_configs.py."""
import os
import random
import shutil

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from circom_amd.circuits.sha256 import Sha256
from circom_amd.hip_elements import bitblast as BB, bitjit as BJ
from oracle.jit_eval import run_ir, JitHazard
from oracle.tape_eval import eval_flat, check_r1cs

from test_bitplane import BitGadget, BitAssert, BadBit, BadWeighted, BadWords


def _rows(fc, n, seed):
    r = random.Random(seed)
    return [[r.randrange(2) for _ in range(fc.n_main_inputs)] for _ in range(n)]


def _run(jp, fc, rows):
    W = len(rows)
    mem = {0: 0, 1: (1 << W) - 1}
    for k in range(fc.n_main_inputs):
        mem[BJ.IN_BASE + k] = sum((rows[i][k] & 1) << i for i in range(W))
    return run_ir(jp, mem, W)


def _flat(fc, row):
    return eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start + k: v for k, v in enumerate(row)})


def _check_witnesses(jp, fc, rows, mem):
    for i, row in enumerate(rows):
        sig, failed = _flat(fc, row)
        assert failed is None
        got = [(mem[int(jp.sig_slot[s])] >> i) & 1 for s in range(fc.n_signals)]
        assert got == sig, i


@pytest.mark.parametrize("nv,na,pf", [(256, 256, 384), (24, 8, 16), (12, 0, 4), (16, 4, 0)])
def test_emitted_program_reproduces_the_flat_code(nv, na, pf):
    """generous registers, and register files so small that every level of the hierarchy is exercised: AccVGPR spills,
    scratch rows, late loads, prefetches that are evicted again"""
    fc = flatten(Program(BitGadget(16)))
    net = BB.bitblast(fc)
    jp = BJ.lower_jit(net, fc, n_vgpr=nv, n_agpr=na, prefetch=pf)
    assert jp.check_complete
    rows = _rows(fc, 40, 3)
    mem, fb, bad = _run(jp, fc, rows)
    assert fb == 0 and bad == 0
    _check_witnesses(jp, fc, rows, mem)
    if nv < 32:
        assert jp.stats["scratch_stores"] + jp.stats["agpr_writes"] + jp.stats["late_loads"] > 0
    assert jp.n_slots * BJ.ROW_BYTES < 1 << 32 and jp.n_slots % 16 == 0


def test_sha256_block_through_the_emitted_program():
    import hashlib
    fc = flatten(Program(Sha256(64)))
    net = BB.bitblast(fc)
    jp = BJ.lower_jit(net, fc)
    assert jp.check_complete and jp.stats["check_unchecked"] == 0 and jp.stats["check_int"] > 0 and jp.stats["check_lut"] > 0
    # a register file of 253 + 256 keeps the block (almost) away from scratch rows, and the two-distance prefetch from stalls
    assert jp.stats["scratch_stores"] < jp.stats["stores"] // 50 and jp.stats["late_loads"] < jp.stats["gates"] // 1000
    rng = random.Random(1)
    msgs = [bytes(rng.randrange(256) for _ in range(8)) for _ in range(33)]
    rows = [[(m[k // 8] >> (7 - k % 8)) & 1 for k in range(64)] for m in msgs]
    mem, fb, bad = _run(jp, fc, rows)
    assert fb == 0 and bad == 0
    for i, m in enumerate(msgs):
        dg = hashlib.sha256(m).digest()
        assert [(mem[int(jp.sig_slot[1 + k])] >> i) & 1 for k in range(256)] == [(dg[k // 8] >> (7 - k % 8)) & 1 for k in range(256)]
    sig, _ = _flat(fc, rows[7])
    assert [(mem[int(jp.sig_slot[s])] >> 7) & 1 for s in range(fc.n_signals)] == sig


@pytest.mark.parametrize("prog", [Program(BadBit(12)), Program(BadWeighted(12, False)), Program(BadWords(13, False)),
                                  Program(BadWords(31, True)), Program(BadWords(0, False))])
def test_fused_check_flags_exactly_the_violating_instances(prog):
    """circuits whose witness code disagrees with a constraint for some inputs: the gates the emitter adds for the constraints
    (LUT class / integer class) must fire for those instances and for no other"""
    fc = flatten(prog)
    net = BB.bitblast(fc)
    for nv, na, pf in ((256, 256, 384), (20, 6, 8)):
        jp = BJ.lower_jit(net, fc, n_vgpr=nv, n_agpr=na, prefetch=pf)
        assert jp.check_complete
        rows = _rows(fc, 64, 11)
        for i in range(0, 64, 3):
            rows[i] = [0] * fc.n_main_inputs           # (BadBit is violated by every instance with a 1 among its inputs)
        mem, fb, bad = _run(jp, fc, rows)
        n_bad = 0
        for i, row in enumerate(rows):
            sig, _ = _flat(fc, row)
            want = check_r1cs(fc.fp.q, fc.constraints, sig) is not None
            assert bool((bad >> i) & 1) == want, i
            n_bad += want
        assert 0 < n_bad < 64


@pytest.mark.parametrize("prog", [Program(BadBit(12)), Program(BadWeighted(12, False)), Program(BadWords(31, True)), Program(Sha256(64))])
def test_audit_program_rechecks_the_table_the_evaluation_left(prog):
    """lower_jit(audit_of=): the check's gates alone, their wires LOADED from the rows the evaluation stored.  On the table
    as the evaluation left it the audit raises the same flags as the fused check; on a table somebody changed afterwards
    (cw_device_bits hands out the raw pointer) it flags exactly the instances whose witness no longer satisfies the system"""
    fc = flatten(prog)
    net = BB.bitblast(fc)
    big = fc.n_signals > 10000
    for nv, na, pf in (((256, 256, 384),) if big else ((256, 256, 384), (20, 6, 8))):
        jp = BJ.lower_jit(net, fc, n_vgpr=nv, n_agpr=na, prefetch=pf)
        ja = BJ.lower_jit(net, fc, n_vgpr=nv, n_agpr=na, prefetch=pf, audit_of=jp)
        assert ja is not None and ja.is_audit and ja.n_slots >= jp.n_slots
        assert not any(ins[0] in ("st", "sta") and ins[2] < jp.n_slots for ins in ja.ir)      # stores: scratch behind the table only
        assert ja.stats["gates"] == jp.stats["check_gates"] or ja.stats["gates"] <= jp.stats["check_gates"]
        W = 24 if big else 64
        rows = _rows(fc, W, 11)
        for i in range(0, W, 3):
            rows[i] = [0] * fc.n_main_inputs
        mem, fb, bad = _run(jp, fc, rows)
        table = {k: v for k, v in mem.items() if k < jp.n_slots}
        ja_run = type("J", (), {"ir": ja.ir, "n_slots": ja.n_slots, "n_vgpr": ja.n_vgpr, "n_agpr": ja.n_agpr})
        _, fb2, bad2 = run_ir(ja_run, dict(table), W)
        assert fb2 == 0 and bad2 == bad
        # flip one stored wire in a few instances: the audit must flag exactly the instances that now violate a constraint
        rng = random.Random(5)
        victim = None
        cand = [s_ for s_ in range(1, fc.n_signals) if not fc.main_input_start <= s_ < fc.main_input_start + fc.n_main_inputs]
        for s_ in rng.sample(cand, min(40, len(cand))):
            if any(s_ in A or s_ in B_ or s_ in C for A, B_, C in fc.constraints):
                victim = s_
                break
        assert victim is not None
        flip = sum(1 << i for i in (1, 5, W - 2))
        t2 = dict(table)
        t2[int(jp.sig_slot[victim])] ^= flip
        _, _, bad3 = run_ir(ja_run, t2, W)
        for i in range(W):
            sig = [(t2[int(jp.sig_slot[s])] >> i) & 1 for s in range(fc.n_signals)]
            assert bool((bad3 >> i) & 1) == (check_r1cs(fc.fp.q, fc.constraints, sig) is not None), i


@template
def WideRow(c, n):
    """a long linear row with field-sized weights over 2 n distinct wires (no exact integer comparison possible)"""
    a = c.input("a", n)
    b = c.input("b", n)
    out = c.output("out", n)
    lhs = c.const(0)
    rhs = c.const(0)
    for k in range(n):
        c.hint(out[k], a[k] + b[k] - 2 * a[k] * b[k])
        w = (1 << 200) * 3 ** k
        lhs = lhs + out[k] * w
        rhs = rhs + a[k] * w
    c.enforce(lhs, rhs, runtime_check=False)


def test_field_sized_rows_are_left_to_the_audit_kernels():
    fc = flatten(Program(WideRow(8)))
    jp = BJ.lower_jit(BB.bitblast(fc), fc)
    assert not jp.check_complete and jp.stats["check_unchecked"] == 1
    # ... while the same weights over few wires are a truth table like any other
    fc = flatten(Program(BadWeighted(12, True)))
    jp = BJ.lower_jit(BB.bitblast(fc), fc)
    assert jp.check_complete and jp.stats["check_lut"] >= 1


def test_assertion_gates_reach_the_fallback_mask():
    fc = flatten(Program(BitAssert()))
    net = BB.bitblast(fc)
    assert net.asserts
    jp = BJ.lower_jit(net, fc)
    rows = [[i & 1, (i >> 1) & 1] for i in range(16)]
    mem, fb, bad = _run(jp, fc, rows)
    assert [(fb >> i) & 1 for i in range(16)] == [int(a != b) for a, b in rows]


def test_replay_rejects_broken_programs():
    """the executor's rules are enforced: a missing wait, a clobbered register, a row written twice"""
    fc = flatten(Program(BitGadget(8)))
    net = BB.bitblast(fc)
    jp = BJ.lower_jit(net, fc, n_vgpr=16, n_agpr=4, prefetch=8)
    rows = _rows(fc, 8, 5)
    _run(jp, fc, rows)
    good = list(jp.ir)
    jp.ir = [ins for ins in good if ins[0] != "w"]            # no waits at all: the first use of a loaded value is a race
    with pytest.raises(JitHazard):
        _run(jp, fc, rows)
    k = next(i for i, ins in enumerate(good) if ins[0] == "st")
    jp.ir = good[:k + 1] + [good[k]] + good[k + 1:]
    with pytest.raises(JitHazard):
        _run(jp, fc, rows)
    k = next(i for i, ins in enumerate(good) if ins[0] == "ld")
    jp.ir = good[:k + 1] + [("g", good[k][1], -1, -1, -1, 0)] + good[k + 1:]
    with pytest.raises(JitHazard):
        _run(jp, fc, rows)


def _eval_gate_text(line, regs):
    """value a printed gate line leaves in its destination, on 32-bit registers `regs` (name -> int)"""
    tok = line.replace(",", " ").split()
    op, dst, srcs = tok[0], tok[1], tok[2:]

    def val(x):
        return 0 if x == "0" else 0xFFFFFFFF if x == "-1" else regs[x]
    if op == "v_bitop3_b32":
        tt = int(srcs[3].split(":")[1], 16)
        a, b, c = (val(x) for x in srcs[:3])
        out = 0
        for i in range(32):
            idx = ((a >> i) & 1) << 2 | ((b >> i) & 1) << 1 | ((c >> i) & 1)
            out |= ((tt >> idx) & 1) << i
        return dst, out
    a, b = val(srcs[0]), val(srcs[1])
    return dst, {"v_and_b32": a & b, "v_or_b32": a | b, "v_xor_b32": a ^ b, "v_xnor_b32": ~(a ^ b) & 0xFFFFFFFF}[op]


def test_every_printed_gate_form_computes_its_table():
    """gate_asm prints a gate with one constant operand as a 4-byte VOP2 instruction when the remaining function is AND / OR /
    XOR / XNOR: every table x every placement of constants / repeated registers, the printed text evaluated on random registers
    against the table itself"""
    r = random.Random(5)
    regs = {"v%d" % i: r.getrandbits(32) for i in range(3, 9)}
    short = 0
    for tt in range(256):
        for ops in ((3, 4, 5), (-1, 4, 5), (3, -1, 5), (3, 4, -1), (-2, 4, 5), (3, -2, 5), (3, 4, -2), (-1, -1, 5), (-2, 4, -1),
                    (3, 3, 5), (-1, 4, 4), (3, -2, 3), (-1, -2, -1)):
            line = BJ.gate_asm(("g", 8, ops[0], ops[1], ops[2], tt))
            short += not line.lstrip().startswith("v_bitop3")
            dst, got = _eval_gate_text(line, regs)
            a, b, c = (0 if x == -1 else 0xFFFFFFFF if x == -2 else regs["v%d" % x] for x in ops)
            want = 0
            for i in range(32):
                want |= ((tt >> (((a >> i) & 1) << 2 | ((b >> i) & 1) << 1 | ((c >> i) & 1))) & 1) << i
            assert dst == "v8" and got == want, (tt, ops, line)
    assert short > 300                                                      # 4 of 16 two-input functions x 6 placements x 16 don't-care fillings


def test_assembly_and_tape_section(tmp_path):
    """the text assembles with the ROCm LLVM tools into a code object that names the kernel; the .cwt carries it and the
    C-ABI loader accepts it (and refuses a damaged section)"""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        pytest.skip("no ROCm assembler here")
    from circom_amd import runtime as rt
    cp = compile_program(Program(BitGadget(8)), str(tmp_path), "bg8j", sym=False, strands=(1,), bits=True, jit=True)
    assert cp.jit is not None and cp.jit.code[:4] == b"\x7fELF" and BJ.KERNEL_NAME.encode() in cp.jit.code
    asm = BJ.to_asm(cp.jit)
    n_acc = sum(1 for i in cp.jit.ir if i[0] == "acc")                      # `acc` rows print as v_or_b32 too
    forms = {f: asm.count(f + " ") for f in ("v_bitop3_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_xnor_b32")}
    assert sum(forms.values()) - n_acc == cp.jit.stats["gates"] and "s_endpgm" in asm
    assert forms["v_bitop3_b32"] < cp.jit.stats["gates"]                    # some gates took the 4-byte forms (and assembled)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert c.bits_info()
    c.close()
    raw = bytearray(open(cp.tape_path, "rb").read())
    at = raw.rfind(b"\x7fELF")
    bad = tmp_path / "bad.cwt"
    for mutate in (lambda b: b.__setitem__(slice(at, at + 4), b"\x7fELG"),          # not a code object
                   lambda b: b.__delitem__(slice(len(b) - 64, len(b)))):            # truncated
        b2 = bytearray(raw)
        mutate(b2)
        bad.write_bytes(bytes(b2))
        with pytest.raises(rt.CwError):
            rt.Circuit(str(bad), cp.dat_path, cp.r1cs_path)
    # a circuit compiled without the emitter still loads (one bit program)
    cp2 = compile_program(Program(BitGadget(8)), str(tmp_path), "bg8i", sym=False, strands=(1,), bits=True, jit=False)
    assert cp2.jit is None
    rt.Circuit(cp2.tape_path, cp2.dat_path, cp2.r1cs_path).close()


def test_chunk_stride_baked_into_both_code_objects_is_the_header_row_count(tmp_path, monkeypatch):
    """the chunk stride is an immediate of the emitted code (rows per chunk x 256 bytes in s10).  The audit's
    scratch rows extend the chunk, so BOTH programs must be printed with the raised row count - the one the tape header carries
    and every other kernel that walks the table uses.  Register-starved lowerings make the audit of a small circuit spill."""
    import functools
    import re
    from circom_amd import compiler
    fc = flatten(Program(BitGadget(16)))
    net = BB.bitblast(fc)
    real_lower, real_asm = BJ.lower_jit, BJ.to_asm
    lowered, strides = [], []

    def starved(net_, fc_, **kw):
        main = "audit_of" not in kw
        # the audit gets fewer registers than the program whose table it reads: its scratch rows are what raises the row count
        jp = real_lower(net_, fc_, n_vgpr=24 if main else 12, n_agpr=8 if main else 0, prefetch=16 if main else 4, **kw)
        lowered.append(jp)
        return jp

    def spy(jp):
        text = real_asm(jp)
        strides.append((jp.is_audit, int(re.search(r"s_mov_b32 s10, (0x[0-9a-f]+)", text).group(1), 16)))
        return text
    monkeypatch.setattr(BJ, "lower_jit", starved)
    monkeypatch.setattr(BJ, "to_asm", spy)
    monkeypatch.setattr(BJ, "assemble", lambda text: b"\x7fELF" + text.encode()[:64])
    jp = compiler.emit_jit(net, fc, True)
    main_rows, audit_rows = (real_lower(net, fc, n_vgpr=24, n_agpr=8, prefetch=16).n_slots, lowered[1].stats["slots"])
    assert audit_rows > main_rows, "the audit of this lowering was meant to spill scratch rows"
    assert jp.n_slots == audit_rows
    assert sorted(strides) == [(False, jp.n_slots * BJ.ROW_BYTES), (True, jp.n_slots * BJ.ROW_BYTES)]
    assert jp.code_stride == jp.audit_stride == jp.n_slots * BJ.ROW_BYTES
    # ... and the writer refuses a program whose code was printed with another stride than its header says
    from circom_amd.hip_elements import writers
    from circom_amd.hip_elements.lower import lower
    from circom_amd.compiler import lower_bitplane_net
    bt, _ = lower_bitplane_net(fc, True)
    jp.code_stride -= BJ.ROW_BYTES * 16
    with pytest.raises(AssertionError, match="code_stride"):
        writers.write_tape(str(tmp_path / "x.cwt"), [lower(fc, n_strands=1, mont=False)], bt, jp, ())


# ---- one body per repeated template: loops -------------------------------------------------------------------------------------
@template
def MixBlock(c, n):
    """a small stand-in for a compression block: state `hin`, message `inp` -> new state; adders, xor3 / maj cells, constants"""
    hin = c.input("hin", n)
    inp = c.input("inp", n)
    out = c.output("out", n)
    x3 = c.component("x3", Xor3(n))
    mj = c.component("mj", Maj_t(n))
    for k in range(n):
        c.set(x3["a"][k], hin[k]); c.set(x3["b"][k], inp[(k + 3) % n]); c.set(x3["c"][k], hin[(k + 5) % n])
        c.set(mj["a"][k], inp[k]); c.set(mj["b"][k], hin[(k + 1) % n]); c.set(mj["c"][k], inp[(k + 7) % n])
    s1 = c.component("s1", BinSum(n, 3))
    for k in range(n):
        c.set(s1["in"][0][k], x3["out"][k]); c.set(s1["in"][1][k], mj["out"][k]); c.set(s1["in"][2][k], (0x5A5A5A5A >> k) & 1)
    s2 = c.component("s2", BinSum(n, 2))
    for k in range(n):
        c.set(s2["in"][0][k], s1["out"][k]); c.set(s2["in"][1][k], hin[k])
    for k in range(n):
        c.set(out[k], s2["out"][k])


@template
def MixChain(c, n, blocks):
    """`blocks` instances of one template in a chain: the first reads constants where the others read their predecessor, the
    last reads constants where the others read inputs - the shape of circomlib's Sha256 (IV, padding)"""
    msg = c.input("msg", n * (blocks - 1) + n // 2)
    out = c.output("out", n)
    prev = None
    for i in range(blocks):
        b = c.component("blk", MixBlock(n), i)
        for k in range(n):
            c.set(b["hin"][k], ((0x6A09E667 >> k) & 1) if prev is None else prev["out"][(k + 1) % n])
            j = i * n + k
            c.set(b["inp"][k], msg[j] if j < n * (blocks - 1) + n // 2 else (k & 1))
        prev = b
    for k in range(n):
        c.set(out[k], prev["out"][k])


from circom_amd.circuits.sha256 import Xor3, Maj_t, BinSum  # noqa: E402


@pytest.mark.parametrize("nv,na,pf", [(256, 256, 384), (28, 8, 16), (14, 0, 4)])
def test_repeated_template_becomes_one_looped_body(nv, na, pf, monkeypatch):
    """ONE body per repeated template (template.rs:160-474): the instances' input signals become ports, the gate lists of the
    instances are isomorphic, the emitter prints one of them inside a loop with per-iteration rows and a table of the rows its
    ports read.  The replay (registers poisoned at every iteration boundary) reproduces the flat code for every signal, the
    fused check and the looped audit flag exactly the violating instances, under generous and starved register files."""
    monkeypatch.setattr(BJ, "OPAQUE_MIN_SIGNALS", 40)
    monkeypatch.setattr(BJ, "LOOP_MIN_BODY_GATES", 16)
    fc = flatten(Program(MixChain(16, 5)))
    ports, marks = BJ.instance_ports(fc)
    assert ports and len(ports) >= 5             # (at this threshold the blocks' sub-components are candidates too: nested ports)
    net = BB.bitblast(fc, ports=ports, marks=marks)
    assert len(net.port_src) >= 5 * 32
    jp = BJ.lower_jit(net, fc, n_vgpr=nv, n_agpr=na, prefetch=pf)
    assert jp.loop and jp.loop["K"] == 5 and jp.stats["loop"]["iterations"] == 5
    straight = BJ.lower_jit(net, fc, n_vgpr=nv, n_agpr=na, prefetch=pf, loop=False)
    assert straight.loop is None and jp.stats["instructions"] < straight.stats["instructions"] / 3
    assert jp.stats["executed"]["gates"] == straight.stats["gates"]
    rows = _rows(fc, 48, 13)
    mem, fb, bad = _run(jp, fc, rows)
    assert fb == 0 and bad == 0
    _check_witnesses(jp, fc, rows, mem)
    mem_s, _, _ = _run(straight, fc, rows)
    _check_witnesses(straight, fc, rows, mem_s)
    # the audit of the looped program's table: looped as well, its scratch inside the spare rows of every iteration
    ja = BJ.lower_jit(net, fc, n_vgpr=nv, n_agpr=na, prefetch=pf, audit_of=jp)
    assert ja is not None and any(i[0] == "loop" for i in ja.ir) and ja.n_slots >= jp.n_slots
    table = {k: v for k, v in mem.items() if k < jp.n_slots}
    ja_run = type("J", (), {"ir": ja.ir, "n_slots": ja.n_slots, "n_vgpr": ja.n_vgpr, "n_agpr": ja.n_agpr})
    _, fb2, bad2 = run_ir(ja_run, dict(table), 48)
    assert fb2 == 0 and bad2 == 0
    rng = random.Random(3)
    cand = [s_ for s_ in range(1, fc.n_signals) if not fc.main_input_start <= s_ < fc.main_input_start + fc.n_main_inputs
            and any(s_ in A or s_ in B_ or s_ in C for A, B_, C in fc.constraints)]
    for victim in rng.sample(cand, 6):
        t2 = dict(table)
        t2[int(jp.sig_slot[victim])] ^= sum(1 << i for i in (0, 17, 46))
        _, _, bad3 = run_ir(ja_run, t2, 48)
        for i in range(48):
            sig = [(t2[int(jp.sig_slot[s])] >> i) & 1 for s in range(fc.n_signals)]
            assert bool((bad3 >> i) & 1) == (check_r1cs(fc.fp.q, fc.constraints, sig) is not None), (victim, i)
    # both programs assemble; the looped one carries its row tables behind the code
    if nv == 256 and os.path.exists("/opt/rocm/lib/llvm/bin/clang"):      # (the starved register files are not launchable shapes)
        text = BJ.to_asm(jp)
        assert "cw_ext_tab0:" in text and "s_setpc_b64" in text and text.count("s_load_dwordx8") >= 1
        assert BJ.assemble(text)[:4] == b"\x7fELF" and BJ.assemble(BJ.to_asm(ja))[:4] == b"\x7fELF"


def test_instances_wired_differently_share_no_loop_when_their_gates_differ(monkeypatch):
    """the loop is only formed over instances whose ordered gate lists are isomorphic: a chain whose middle instance is a
    different template instance (other parameter) keeps straight-line code - and still evaluates correctly through its ports"""
    monkeypatch.setattr(BJ, "OPAQUE_MIN_SIGNALS", 40)
    monkeypatch.setattr(BJ, "LOOP_MIN_BODY_GATES", 16)

    @template
    def Odd(c):
        msg = c.input("msg", 40)
        out = c.output("out", 16)
        a = c.component("a", MixBlock(16))
        b = c.component("b", MixBlock(16))
        for k in range(16):
            c.set(a["hin"][k], msg[k]); c.set(a["inp"][k], msg[16 + k])
        for k in range(16):
            c.set(b["hin"][k], a["out"][k]); c.set(b["inp"][k], a["out"][(k + 2) % 16] if k < 8 else msg[32 + k - 8])
        for k in range(16):
            c.set(out[k], b["out"][k])
    fc = flatten(Program(Odd()))
    ports, marks = BJ.instance_ports(fc)
    net = BB.bitblast(fc, ports=ports, marks=marks)
    jp = BJ.lower_jit(net, fc)
    rows = _rows(fc, 32, 4)
    mem, fb, bad = _run(jp, fc, rows)
    assert fb == 0 and bad == 0
    _check_witnesses(jp, fc, rows, mem)          # (two isomorphic instances: a loop of two, or none - either way exact)


def test_check_rows_drop_a_wire_that_a_checked_product_row_pins():
    """build_check: `mid <== b * c` is a checked row of its own, so the `out` rows of Xor3 / Maj are checked with mid replaced by
    b & c (5 -> 4 wires: fewer gates, and mid need not be kept for them).  Instance-level verdicts must not change: the table an
    evaluation left is clean, and after flipping ANY single wire - the pinned mid, an operand, the output - in some instances the
    audit flags exactly the instances that violate a constraint (the product row raises the flag when mid itself is wrong)."""
    @template
    def Cells(c, n):
        a = c.input("a", n); b = c.input("b", n); cc = c.input("c", n)
        out = c.output("out", n)
        x3 = c.component("x3", Xor3(n)); mj = c.component("mj", Maj_t(n))
        for k in range(n):
            c.set(x3["a"][k], a[k]); c.set(x3["b"][k], b[k]); c.set(x3["c"][k], cc[k])
        for k in range(n):
            c.set(mj["a"][k], x3["out"][k]); c.set(mj["b"][k], b[(k + 1) % n]); c.set(mj["c"][k], cc[k])
        for k in range(n):
            c.set(out[k], mj["out"][k])
    fc = flatten(Program(Cells(6)))
    net = BB.bitblast(fc)
    with_sub = BJ.lower_jit(net, fc)
    assert with_sub.stats["check_substituted"] == 12                      # the out rows of 6 Xor3 and 6 Maj cells
    BJ.SUBSTITUTE_PRODUCTS = False
    try:
        net._check_cache = None
        without = BJ.lower_jit(net, fc)
    finally:
        BJ.SUBSTITUTE_PRODUCTS = True
        net._check_cache = None
    assert with_sub.stats["check_gates"] < without.stats["check_gates"]
    jp = BJ.lower_jit(net, fc)
    ja = BJ.lower_jit(net, fc, audit_of=jp)
    W = 40
    rows = _rows(fc, W, 2)
    mem, fb, bad = _run(jp, fc, rows)
    assert fb == 0 and bad == 0
    _check_witnesses(jp, fc, rows, mem)
    table = {k: v for k, v in mem.items() if k < jp.n_slots}
    ja_run = type("J", (), {"ir": ja.ir, "n_slots": ja.n_slots, "n_vgpr": ja.n_vgpr, "n_agpr": ja.n_agpr})
    names = fc.signal_names()
    flipped_mid = 0
    for s_ in range(1, fc.n_signals):
        t2 = dict(table)
        t2[int(jp.sig_slot[s_])] ^= sum(1 << i for i in (0, 13, W - 1))
        _, _, bad3 = run_ir(ja_run, t2, W)
        flipped_mid += ".mid[" in names[s_]
        for i in range(W):
            sig = [(t2[int(jp.sig_slot[s])] >> i) & 1 for s in range(fc.n_signals)]
            assert bool((bad3 >> i) & 1) == (check_r1cs(fc.fp.q, fc.constraints, sig) is not None), (names[s_], i)
    assert flipped_mid == 12
