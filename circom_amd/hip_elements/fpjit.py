"""hip_elements, emitted 256-bit code, part 2: the row stream of a schedule as STRAIGHT-LINE gfx950 code.

The reference's C back-end turns the witness program into code (one C++ function per template,
compiler/src/circuit_design/template.rs:174-474; a `Fr_mul(&dst, &a, &b)` call per ComputeBucket,
compute_bucket.rs:315-421); the 256-bit engine of rounds 1-3 INTERPRETS its schedule instead: `cw_eval_kernel` spends
~300 instructions of glue per row (a 38-way switch, operand-kind tests, descriptor loads, conservative `s_waitcnt`s that
also wait for the previous row's stores) around ~320 instructions of arithmetic.  This module is the emitting
counterpart: a lowered schedule (`lower.Tape`: the rows of every strand with their operand kinds, extra destinations, term
tables, barriers - exactly what the interpreter walks and what `oracle/tape_eval.py` replays and race-checks) is printed as
one kernel in which every row is

    [address arithmetic + loads of the NEXT step's operands]      prefetch, one step ahead, into the other register set
    s_waitcnt vmcnt(N)                                             N counted exactly: the stores issued since are not waited for
    [constants as literals, PREV as register moves]
    s_swappc_b64 -> row body (fpjit_bodies.py)                     the operator, compiled from the interpreter's own C++
    [address arithmetic + stores of the result]

so the glue is ~20 instructions, nothing is decoded at run time, and a wave of strand s runs strand s's own code.  The
execution model is the interpreter's (one-step-ahead prefetch, PREV forwarding, LDS hand-off slots, LIGHT/FULL barriers),
so the schedule's race analysis carries over; what is new - wait counts, register sets, clobbers - is replayed on the CPU
by `oracle/fpjit_eval.py` from the IR this module returns next to the text.

Value table, constants, status words, Montgomery forms: unchanged (`csrc/cw_kernels.hip` ingest / check / egress kernels
serve both engines).  LDS hand-off slots are 2 KiB here whatever the lane count (literal offsets).
"""
from __future__ import annotations

import os

import numpy as np

from . import fpjit_bodies as FB
from .lower import (D_COPY, D_ADD, D_SUB, D_NEG, D_MMUL, D_INV, D_IDIV, D_MOD, D_POW, D_SHL, D_SHR, D_BAND, D_BOR, D_BXOR,
                    D_BNOT, D_LT, D_GT, D_LEQ, D_GEQ, D_EQ, D_NEQ, D_LAND, D_LOR, D_LNOT, D_SELECT, D_EXT, D_ASSERT_EQ,
                    D_ASSERT_NZ, D_BARRIER, D_MUL2, D_MADD, D_MULC, D_MADDC, D_LINSUM, D_BIT, D_DOTC, D_CALL,
                    SH_DK, SH_AK, SH_BK, SH_NX, SH_FLAG, X_TMP, X_LDS, K_LDS, KD_NONE, KO_PREV, D_BITS, X_NEXT, D_ALSO)

KERNEL_NAME = "cw_fp_jit"
K_SIG, K_TMP, K_CONST = 0, 1, 2
LDS_SLOT = 2048
PARK_STRAND = 768             # per strand: status words | fused-check findings | address of the kernel's argument block
PARK_BYTES = 16 * PARK_STRAND             # start of the workgroup's LDS: 512 bytes per strand where the caller of a heavy body parks the status
                              # word and the fused check's finding (the body has no register to keep them in)
KERNARG_BYTES = 32 + 4 * FB.N_PARAM_SGPRS + 4 + 24     # V, status, Bp, batch, lanes, pad, FpParams (53 dwords), pad, then the
assert KERNARG_BYTES == 8 * (FB.KA_FTAB + 1)           # tables of tier 2: constants, function bytecode, function table -> 272
VMCNT_MAX, LGKM_MAX = 63, 15

# owned registers (fpjit_bodies.py keeps them live through every body)
V_VLO, V_VHI, V_L16, V_FB, V_ST, V_L16B, V_L16C, V_I = 120, 121, 122, 123, 124, 125, 126, 127
S_BATCH, S_VBASE, S_STATUS, S_RET, S_STRIDE, S_WGBASE = 93, 94, 96, 98, 100, 101

_TWO = {D_ADD: "add", D_SUB: "sub", D_MMUL: "mmul", D_MUL2: "mul2", D_SHL: "shl", D_SHR: "shr", D_BAND: "band", D_BOR: "bor",
        D_BXOR: "bxor", D_LT: "lt", D_GT: "gt", D_LEQ: "leq", D_GEQ: "geq", D_EQ: "eq", D_NEQ: "neq", D_LAND: "land",
        D_LOR: "lor", D_MADD: "madd", D_EXT: "ext"}
_ONE = {D_NEG: "neg", D_BNOT: "bnot", D_LNOT: "lnot"}
_HEAVY = {D_INV: "inv_h", D_POW: "pow_h", D_IDIV: "idiv_h", D_MOD: "mod_h"}
_NO_VALUE = (D_ASSERT_EQ, D_ASSERT_NZ, D_SELECT)


class FpJitProgram:
    def __init__(self):
        self.n_strands = 1
        self.asm = ""
        self.code = None          # the code object (ELF) after assemble()
        self.ir = None            # per strand: list of IR tuples (oracle/fpjit_eval.py)
        self.lds_bytes = 0
        self.scratch_bytes = 0
        self.covered = []         # per constraint: 1 = checked by the emitted code itself (the stand-alone kernel skips it)
        self.stats = {}


def _limbs32(v):
    return [(v >> (32 * k)) & 0xFFFFFFFF for k in range(8)]


def _limbs29(v):
    return [(v >> (29 * k)) & 0x1FFFFFFF for k in range(9)]


class _Step:
    """one call of a body (or an inline operation) with what it needs from memory and what it leaves there"""
    __slots__ = ("loads", "pre", "body", "inline", "stores", "barrier", "heavy", "value", "sarg", "coef", "check", "call")

    def __init__(self):
        self.loads = []        # (which 'A' | 'B', kind, index)  kind: K_SIG / K_TMP (unified slot = index) or K_LDS
        self.pre = []          # ("prev", which) | ("const", which | 'G', value) | ("zero", 'G') | ("zacc", n)
        self.body = None       # body name without parity suffix ("mmul"), or full name for parity-less bodies
        self.inline = None     # ("bit", k) | ("copy",)
        self.stores = []       # (kind, index)
        self.barrier = None    # None | 0 (LIGHT) | 1 (FULL): the step IS a barrier
        self.heavy = False
        self.value = False     # the step leaves a new value in D
        self.sarg = None       # 64-bit literal for s[36:37]
        self.coef = None       # nine 29-bit limbs for s[24:32]
        self.check = False     # a step of the fused R1CS check (plan_checks)
        self.call = None       # D_CALL: (function id, first slot of the register window (unified), index of the flat operation)


def expand_steps(tape, strand):
    """rows of one strand -> steps.  Slots are unified: signal s -> s, temp slot t -> n_signals + t."""
    rows = tape.rows
    r0, r1 = int(tape.stream_off[strand]), int(tape.stream_off[strand + 1])
    xp = int(tape.extra_off[strand])
    tp = int(tape.term_off[strand])
    sp = int(tape.seq_off[strand]) if len(tape.seq_off) > strand else 0
    ns = tape.n_signals
    consts = tape.consts
    steps = []

    def opnd(st, which, k, v):
        if k == KO_PREV:
            st.pre.append(("prev", which))
        elif k == K_CONST:
            st.pre.append(("const", which, consts[v]))
        elif k == K_LDS:
            st.loads.append((which, K_LDS, v))
        elif k == K_SIG:
            st.loads.append((which, K_SIG, v))
        elif k == K_TMP:
            st.loads.append((which, K_SIG, ns + v))
        else:
            raise ValueError("operand kind %d" % k)

    for r in range(r0, r1):
        w0, dst, a_, b_ = (int(x) for x in rows[r])
        op, dk, ak, bk = w0 & 0xFF, (w0 >> SH_DK) & 7, (w0 >> SH_AK) & 7, (w0 >> SH_BK) & 7
        nx, flag = (w0 >> SH_NX) & 0xFFF, (w0 >> SH_FLAG) & 3
        if op == D_BARRIER:
            st = _Step()
            st.barrier = 1 if dst == 1 else 0
            steps.append(st)
            continue
        seq = None
        if op in (D_ASSERT_EQ, D_ASSERT_NZ, D_IDIV, D_MOD, D_CALL):
            seq = int(tape.seqs[sp])
            sp += 1
        if op == D_ALSO:
            # the spacer behind a D_BITS row (lower.py): the step behind it requests its operands one step ahead - with the spacer in
            # between that is after the bits were stored
            st = _Step()
            st.inline = ("nop",)
            xp += nx
            steps.append(st)
            continue
        if op == D_BITS:
            # consecutive bits of one operand, from bit b_: every extra entry is one store of the current bit, an entry flagged
            # X_NEXT moves on to the next bit first (cw_tape.h; the interpreter's case D_BITS).  One step: the operand is loaded
            # once, every entry is a bit extraction and a store (signals: the lower half only - the table starts cleared)
            ents = []
            for e in tape.extras[xp:xp + nx]:
                e = int(e)
                if e & X_LDS:
                    raise ValueError("bit-field destinations live in the value table")
                idx = e & 0x1FFFFFFF
                ents.append((ns + idx if e & X_TMP else idx, bool(e & X_NEXT), not (e & X_TMP)))
            xp += nx
            st = _Step()
            st.inline = ("bits", b_, tuple(ents))
            opnd(st, "A", ak, a_)
            steps.append(st)
            continue
        stores = []
        if op not in _NO_VALUE and op != D_CALL:
            if dk == K_SIG:
                stores.append((K_SIG, dst))
            elif dk == K_TMP:
                stores.append((K_SIG, ns + dst))
            elif dk == K_LDS:
                stores.append((K_LDS, dst))
            for e in tape.extras[xp:xp + nx]:
                e = int(e)
                if e & X_LDS:
                    stores.append((K_LDS, e & 0x3FFFFFFF))
                elif e & X_TMP:
                    stores.append((K_SIG, ns + (e & 0x3FFFFFFF)))
                else:
                    stores.append((K_SIG, e))
        elif nx:
            raise ValueError("extra destinations on a row without a value")
        xp += nx
        st = _Step()
        if op in _TWO:
            st.body = _TWO[op]
            opnd(st, "A", ak, a_)
            opnd(st, "B", bk, b_)
            st.value = True
        elif op in _ONE:
            st.body = _ONE[op]
            opnd(st, "A", ak, a_)
            st.value = True
        elif op in _HEAVY:
            st.body = _HEAVY[op]
            st.heavy = True
            opnd(st, "A", ak, a_)
            if op != D_INV:
                opnd(st, "B", bk, b_)
            if seq is not None:
                st.sarg = seq
            st.value = True
        elif op in (D_MULC, D_MADDC):
            st.body = ("mulc0", "mulcp", "mulcn")[flag] + ("a" if op == D_MADDC else "")
            opnd(st, "A", ak, a_)
            assert bk == K_CONST
            st.pre.append(("const", "B", consts[b_]))
            if flag:
                st.sarg = consts[b_ + 1]
                assert st.sarg < (1 << 63)
            st.value = True
        elif op == D_COPY:
            st.inline = ("copy",)
            opnd(st, "A", ak, a_)
            st.value = True
        elif op == D_BIT:
            st.inline = ("bit", b_)
            opnd(st, "A", ak, a_)
            st.value = True
        elif op == D_SELECT:
            st.body = "select"
            opnd(st, "A", ak, a_)
        elif op == D_ASSERT_EQ:
            st.body = "asserteq"
            opnd(st, "A", ak, a_)
            opnd(st, "B", bk, b_)
            st.sarg = seq
        elif op == D_ASSERT_NZ:
            st.body = "assertnz"
            opnd(st, "A", ak, a_)
            st.sarg = seq
        elif op in (D_LINSUM, D_DOTC):
            n = a_
            c0 = consts[b_] if bk == K_CONST else None
            terms = [tuple(int(x) for x in t) for t in tape.terms[tp:tp + n]]
            tp += n
            first = True
            for j, (tk, tv, lo, hi) in enumerate(terms):
                t = _Step()
                kind = tk & 7
                if first:
                    t.pre.append(("const", "G", c0) if c0 is not None else ("zero", "G"))
                    t.pre.append(("zacc", 12 if op == D_LINSUM else 36))
                    first = False
                opnd(t, "A", kind, tv)
                if op == D_LINSUM:
                    t.body = "linn" if tk >> 31 else "linp"
                    t.sarg = lo | (hi << 32)
                    assert t.sarg < (1 << 63)
                else:
                    t.body = "dotmac"
                    t.coef = _limbs29(tape.lconsts[lo])
                steps.append(t)
                if op == D_DOTC and (j & 3) == 3 and j + 1 < n:
                    t = _Step()
                    t.body = "dotred"
                    steps.append(t)
            st.body = "linfin" if op == D_LINSUM else "dotfin"
            if first:                       # no terms at all: the constant alone
                st.pre.append(("const", "G", c0) if c0 is not None else ("zero", "G"))
                st.pre.append(("zacc", 12 if op == D_LINSUM else 36))
            st.value = True
        elif op == D_CALL:
            # tier 2: the interpreter of csrc/cw_call.hip.h as one (heavy) body; arguments and results live in the call's
            # register window in the value table, where ordinary rows stored / will load them
            assert bk == K_TMP
            # call_h: one wave per workgroup, up to 512 registers, no native long_div; call_k: the strand kernels' 128 VGPRs
            # (spills into a private segment), with it - what several strands or a long_div function need
            st.body = "call_k" if (tape.n_strands > 1 or any(f_[2] is not None and f_[2][0] == 4 for f_ in (tape.functions or ()))) else "call_h"
            st.heavy = True
            st.call = (a_, ns + b_, seq)
        else:
            raise ValueError("device op %d has no emitted form" % op)
        st.stores = stores
        steps.append(st)
    return steps


def plan_checks(tape, constraints, all_steps):
    """The fused R1CS check: every constraint of a class the emitted code knows becomes one or a few steps that recompute
    it from the STORED wires (or from D while the wire just produced is still there) right behind the row that produces its
    last wire, in that row's strand - the wires are in registers or the CU's caches, not in HBM.  A violated constraint
    lowers the instance's `fb` (smallest violated index); the strand's last call publishes it.  Classes: wire equalities,
    single products a * b = c, w1 + k = w2, linear rows (any coefficients; one reduction per four terms, as D_DOTC).
    Constraints with a multi-term factor stay with the stand-alone kernel (`covered[i]` = 0), which then skips the others.
    Returns covered: list of 0/1 per constraint."""
    q, ns, S = tape.q, tape.n_signals, tape.n_strands
    mont = bool(getattr(tape, "mont", False))
    R = pow(2, tape.rbits, q)
    one = R if mont else 1
    # where and when every signal is produced: (strand, step index, epoch); epochs are counted in barriers
    prod = {}
    full_after = []                       # per barrier: is it FULL
    for s_, steps in enumerate(all_steps):
        ep = 0
        for k, st in enumerate(steps):
            if st.barrier is not None:
                if s_ == 0:
                    full_after.append(st.barrier == 1)
                ep += 1
                continue
            for kind, idx in st.stores:
                if kind == K_SIG and idx < ns:
                    prod[idx] = (s_, k, ep)
    # first step index of every epoch, per strand (the step right behind the barrier that closes the previous epoch)
    ep_start = []
    for steps in all_steps:
        starts, ep = {0: 0}, 0
        for k, st in enumerate(steps):
            if st.barrier is not None:
                ep += 1
                starts[ep] = k + 1
        ep_start.append(starts)
    n_ep = len(full_after) + 1

    def next_full(e):                     # first epoch whose rows see, through the table, what another strand stored in epoch e
        for b in range(e, len(full_after)):
            if full_after[b]:
                return b + 1
        return None

    stats = {"same_register": 0}
    plan_checks.stats = stats
    inserts = [dict() for _ in range(S)]  # strand -> position (insert BEFORE step index) -> list of steps
    load = [sum(300 if st.body else 20 for st in steps) for steps in all_steps]      # rough instruction counts per strand
    covered = [0] * len(constraints)

    def sval(v):
        v %= q
        return v - q if v > q // 2 else v

    for ci, (A, B, C) in enumerate(constraints):
        A = {k: v % q for k, v in A.items() if v % q}
        B = {k: v % q for k, v in B.items() if v % q}
        C = {k: v % q for k, v in C.items() if v % q}
        quad = bool(A) and bool(B)
        # ---- classify -------------------------------------------------------------------------------------------------------
        plan = None
        if quad:
            if len(A) == 1 and len(B) == 1 and len(C) == 1:
                (wa, ca), (wb, cb), (wc, cc) = next(iter(A.items())), next(iter(B.items())), next(iter(C.items()))
                if wa and wb and wc and ca * cb % q == cc:
                    plan = ("mul", wa, wb, wc)
            if plan is None and len(A) == 1 and len(B) == 1 and 0 not in A and 0 not in B and len(C) <= 64:
                # a * b = (a linear right-hand side): the row's C part, scaled by 1 / (ca cb), is summed like a linear row
                (wa, ca), (wb, cb) = next(iter(A.items())), next(iter(B.items()))
                sc = pow(ca * cb % q, -1, q)
                k0 = C.get(0, 0) * sc % q
                plan = ("mulc", wa, wb, [(w, v * sc % q) for w, v in sorted(C.items()) if w], k0)
        else:
            # a linear row: sum(terms) = 0.  A == {} or B == {}: the row is C = 0 (A * B vanishes)
            lin = dict(C)
            k0 = lin.pop(0, 0)             # the constant term (times the constant-one wire)
            wires = sorted(lin)
            if len(wires) == 2 and k0 == 0 and (lin[wires[0]] + lin[wires[1]]) % q == 0 and sval(lin[wires[0]]) in (1, -1):
                plan = ("eq", wires[0], wires[1])
            elif len(wires) == 2 and (lin[wires[0]] + lin[wires[1]]) % q == 0 and sval(lin[wires[0]]) in (1, -1):
                # +w1 - w2 + k = 0  ->  w1 + k = w2
                w1, w2 = (wires[0], wires[1]) if sval(lin[wires[0]]) == 1 else (wires[1], wires[0])
                plan = ("add", w1, k0, w2)
            elif 1 <= len(wires) <= 64:
                # sum_i c_i w_i + k = rhs: one wire with coefficient -1 becomes the right-hand side when there is one
                rhs = next((w for w in wires if sval(lin[w]) == -1), None)
                plan = ("dot", [(w, lin[w]) for w in wires if w != rhs], k0, rhs)
        if plan is None:
            continue
        if plan[0] == "eq" and plan[1] in prod and plan[2] in prod and prod[plan[1]][:2] == prod[plan[2]][:2]:
            covered[ci] = 1                 # both wires are stored from D by one and the same row (component wiring): they ARE one
            stats["same_register"] += 1     # register; nothing to recompute
            continue
        ws = [w for w in (plan[1:4] if plan[0] in ("mul", "eq") else [plan[1], plan[3]] if plan[0] == "add"
                          else [plan[1], plan[2]] + [w for w, _ in plan[3]] if plan[0] == "mulc"
                          else [w for w, _ in plan[1]] + ([plan[3]] if plan[3] is not None else [])) if w]
        # ---- place: wires of other strands must have crossed a FULL barrier; among the strands that can see every wire the
        # least loaded one takes the check (most strands of a strand-parallel schedule wait at barriers most of the time),
        # the producer of the last wire first when loads tie
        last, pos_key = 0, (-1, -1)
        for w in ws:
            if w in prod and (prod[w][2], prod[w][1]) > pos_key:
                pos_key = (prod[w][2], prod[w][1])
                last = prod[w][0]
        best = None
        for cand in sorted(range(S), key=lambda x: (load[x], x != last)):
            pos, ok = 0, True
            for w in ws:
                if w not in prod:
                    continue                              # a main input: there before the kernel starts
                s_, k, ep = prod[w]
                if s_ == cand:
                    pos = max(pos, k + 1)
                else:
                    e2 = next_full(ep)
                    if e2 is None or e2 not in ep_start[cand]:
                        ok = False
                        break
                    pos = max(pos, ep_start[cand][e2])
            if ok:
                best = (cand, pos)
                break
        if best is None:
            continue
        owner, pos = best
        load[owner] += {"eq": 60, "mul": 380, "add": 170}.get(plan[0], 0) or (400 + 180 * len(plan[3] if plan[0] == "mulc" else plan[1]))
        # never inside the term steps of a LINSUM / DOTC row (they own G and the accumulators): move behind the row's end
        steps = all_steps[owner]
        while pos < len(steps) and pos > 0 and steps[pos - 1].body in ("linp", "linn", "dotmac", "dotred") and not steps[pos - 1].check:
            pos += 1
        inserts[owner].setdefault(pos, []).append((ci, plan))
        covered[ci] = 1

    # ---- build the steps ---------------------------------------------------------------------------------------------------------
    for s_ in range(S):
        steps = all_steps[s_]
        out = []
        prev_set = set()                   # slots the latest value step stored: their value is still in D
        for k in range(len(steps) + 1):
            for ci, plan in inserts[s_].get(k, ()):
                def opnd(st, which, w):
                    if w in prev_set:
                        st.pre.append(("prev", which))
                    else:
                        st.loads.append((which, K_SIG, w))

                def new():
                    st = _Step()
                    st.check = True
                    return st

                if plan[0] == "eq" and plan[1] in prev_set and plan[2] in prev_set:
                    stats["same_register"] += 1           # both wires were stored from D by the same row: nothing to compare
                    continue
                if plan[0] == "eq":
                    st = new()
                    st.body = "chkeq"
                    opnd(st, "A", plan[1]); opnd(st, "B", plan[2])
                    st.sarg = ci
                    out.append(st)
                elif plan[0] in ("mul", "add"):
                    st0 = new()                           # the right-hand side goes to G one step earlier (G cannot be prefetched into)
                    st0.inline = ("stashg",)
                    opnd(st0, "A", plan[3])
                    out.append(st0)
                    st = new()
                    if plan[0] == "mul":
                        st.body = "chkmul" if mont else "chkmul2"
                        opnd(st, "A", plan[1]); opnd(st, "B", plan[2])
                    else:
                        st.body = "chkadd"
                        opnd(st, "A", plan[1])
                        st.pre.append(("const", "B", plan[2] * one % q))
                    st.sarg = ci
                    out.append(st)
                else:
                    if plan[0] == "mulc":
                        terms, k0, rhs = plan[3], plan[4], None
                    else:
                        terms, k0, rhs = plan[1], plan[2], plan[3]
                    first = True
                    for j, (w, cf) in enumerate(terms):
                        t = new()
                        if first:
                            t.pre.append(("const", "G", k0 * one % q))
                            t.pre.append(("zacc", 36))
                            first = False
                        opnd(t, "A", w)
                        t.body = "dotmac"
                        t.coef = _limbs29(cf * R % q)
                        out.append(t)
                        if (j & 3) == 3 and j + 1 < len(terms):
                            t = new()
                            t.body = "dotred"
                            out.append(t)
                    st = new()
                    if first:
                        st.pre.append(("const", "G", k0 * one % q))
                        st.pre.append(("zacc", 36))
                    if plan[0] == "mulc":
                        if not first:                     # G = the right-hand side; then the product against it
                            st.body = "dotred"
                            out.append(st)
                            st = new()
                        st.body = "chkmul" if mont else "chkmul2"
                        opnd(st, "A", plan[1]); opnd(st, "B", plan[2])
                    else:
                        st.body = "chkdot"
                        if rhs is not None:
                            opnd(st, "A", rhs)
                        else:
                            st.pre.append(("const", "A", 0))
                    st.sarg = ci
                    out.append(st)
            if k < len(steps):
                st = steps[k]
                out.append(st)
                if st.value:
                    prev_set = {idx for kind, idx in st.stores if kind == K_SIG}
        all_steps[s_] = out
    return covered


_EXP = os.environ.get("CW_FPJIT_EXP", "")


class _Spool(list):
    """the program text; a circuit of millions of rows does not fit a Python list of lines (the ECDSA verifier: ~80 M lines),
    so beyond a threshold the lines go to a file as they are produced (the last few stay for the emitter's look-behind)"""

    def __init__(self, path=None, limit=2_000_000):
        super().__init__()
        self.path, self.limit, self.f = path, limit, None

    def append(self, x):
        list.append(self, x)
        if self.path and len(self) >= self.limit:
            self._flush(keep=4)

    def _flush(self, keep):
        if self.f is None:
            self.f = open(self.path, "w")
        n = len(self) - keep
        self.f.write("".join(self[:n]))
        del self[:n]

    def finish(self):
        """the whole text as a string, or None when it went to `path`"""
        if self.f is None:
            return "".join(self)
        self._flush(keep=0)
        self.f.close()
        return None


class _NoIR(list):
    def append(self, x):
        pass


class _Emitter:
    def __init__(self, tape, bodies, spool_path=None):
        self.tape, self.bodies = tape, bodies
        self.L = _Spool(spool_path)
        self.keep_ir = spool_path is None
        self.ir = [] if self.keep_ir else _NoIR()
        self.used_bodies = set()
        self.n_call = 0
        self.vm_issued = 0          # vector-memory instructions issued so far by this strand (loads and stores, in order)
        self.lg_issued = 0          # LDS instructions issued so far
        self.acc_zero = 0           # leading ACC registers known to be zero
        self.stats = {"steps": 0, "calls": 0, "glue": 0, "loads": 0, "stores": 0}

    def add(self, s, glue=1):
        # timing experiments (results are garbage, tools/fpjit_timing_exp.sh): which part of a launch is what
        if _EXP and ((_EXP == "nowait" and s.startswith("s_waitcnt vmcnt")) or (_EXP == "nobarrier" and s == "s_barrier") or
                     (_EXP == "nostore" and s.startswith("global_store")) or (_EXP == "nocall" and s.startswith("s_swappc") and "publish" not in self.L[-2])):
            return
        self.L.append("  " + s + "\n")
        self.stats["glue"] += glue

    # ---- addresses -----------------------------------------------------------------------------------------------------
    def table_addr(self, slot, sp):
        """s[sp:sp+1] = V + slot * stride"""
        if slot == 0:
            self.add("s_mov_b64 s[%d:%d], s[%d:%d]" % (sp, sp + 1, S_VBASE, S_VBASE + 1))
            return
        self.add("s_mul_i32 s%d, s%d, 0x%x" % (sp, S_STRIDE, slot))
        self.add("s_mul_hi_u32 s%d, s%d, 0x%x" % (sp + 1, S_STRIDE, slot))
        self.add("s_add_u32 s%d, s%d, s%d" % (sp, sp, S_VBASE))
        self.add("s_addc_u32 s%d, s%d, s%d" % (sp + 1, sp + 1, S_VBASE + 1))

    @staticmethod
    def lds_addr(slot):
        off = PARK_BYTES + slot * LDS_SLOT
        base = (V_L16, V_L16B, V_L16C)[off >> 16]
        return base, off & 0xFFFF

    def issue_load(self, reg, kind, idx, sp):
        if kind == K_LDS:
            base, off = self.lds_addr(idx)
            self.add("ds_read_b128 v[%d:%d], v%d offset:%d" % (reg, reg + 3, base, off))
            self.add("ds_read_b128 v[%d:%d], v%d offset:%d" % (reg + 4, reg + 7, base, off + 1024))
            self.lg_issued += 2
            self.ir.append(("ldl", reg, idx, self.lg_issued))
        else:
            self.table_addr(idx, sp)
            self.add("global_load_dwordx4 v[%d:%d], v%d, s[%d:%d]" % (reg, reg + 3, V_VLO, sp, sp + 1))
            self.add("global_load_dwordx4 v[%d:%d], v%d, s[%d:%d]" % (reg + 4, reg + 7, V_VHI, sp, sp + 1))
            self.vm_issued += 2
            self.ir.append(("ld", reg, idx, self.vm_issued))
        self.stats["loads"] += 1

    def issue_store(self, kind, idx):
        d = FB.D_REG
        if kind == K_LDS:
            base, off = self.lds_addr(idx)
            self.add("ds_write_b128 v%d, v[%d:%d] offset:%d" % (base, d, d + 3, off))
            self.add("ds_write_b128 v%d, v[%d:%d] offset:%d" % (base, d + 4, d + 7, off + 1024))
            self.lg_issued += 2
            self.ir.append(("stl", idx, self.lg_issued))
        else:
            self.table_addr(idx, 4)
            self.add("global_store_dwordx4 v%d, v[%d:%d], s[4:5]" % (V_VLO, d, d + 3))
            self.add("global_store_dwordx4 v%d, v[%d:%d], s[4:5]" % (V_VHI, d + 4, d + 7))
            self.vm_issued += 2
            self.ir.append(("st", idx, self.vm_issued))
        self.stats["stores"] += 1

    def mov_fe(self, dst, src):
        for k in range(8):
            self.add("v_mov_b32 v%d, v%d" % (dst + k, src + k))
        self.ir.append(("mov", dst, src))

    def lit_fe(self, dst, value):
        for k, w in enumerate(_limbs32(value)):
            self.add("v_mov_b32 v%d, 0x%x" % (dst + k, w))
        self.ir.append(("lit", dst, value))

    def call(self, name):
        b = self.bodies[name]
        self.used_bodies.add(name)
        k = self.n_call
        self.n_call += 1
        self.add("s_getpc_b64 s[2:3]")
        self.L.append(".Lpc_%d_%d:\n" % (self.strand, k))
        self.add("s_add_u32 s2, s2, fj_body_%s-.Lpc_%d_%d" % (name, self.strand, k))
        self.add("s_addc_u32 s3, s3, 0")
        self.add("s_swappc_b64 s[%d:%d], s[2:3]" % (S_RET, S_RET + 1))
        self.stats["calls"] += 1
        # which leading ACC registers are still zero afterwards
        w = [r for r in b.vwritten if r >= FB.ACC_REG and r < FB.ACC_REG + 36]
        if name in ("dotred", "dotfin") or name.startswith("chkdot"):
            self.acc_zero = 36
        elif w:
            self.acc_zero = min(self.acc_zero, min(w) - FB.ACC_REG)
        self.ir.append(("call", name))

    def rederive(self):
        """the owned vector registers from the lane number (after a heavy body, and in the prologue)"""
        a = self.add
        a("v_mbcnt_lo_u32_b32 v%d, -1, 0" % V_L16)
        a("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (V_L16, V_L16))          # lane
        a("v_add_u32 v%d, s%d, v%d" % (V_I, S_WGBASE, V_L16))             # instance
        a("v_lshlrev_b32 v%d, 4, v%d" % (V_VLO, V_I))
        a("s_lshr_b32 s4, s%d, 1" % S_STRIDE)                            # Bp * 16
        a("v_add_u32 v%d, s4, v%d" % (V_VHI, V_VLO))
        a("v_lshlrev_b32 v%d, 4, v%d" % (V_L16, V_L16))
        a("v_add_u32 v%d, 0x10000, v%d" % (V_L16B, V_L16))
        a("v_add_u32 v%d, 0x20000, v%d" % (V_L16C, V_L16))

    # ---- one strand ------------------------------------------------------------------------------------------------------
    def strand_code(self, strand, steps, prio, park_off):
        self.strand = strand
        self.vm_issued = self.lg_issued = 0
        self.acc_zero = 0
        self.ir = [] if self.keep_ir else _NoIR()
        a = self.add
        self.L.append("fj_strand_%d:\n" % strand)
        debug = bool(os.environ.get("CW_FPJIT_DEBUG"))
        if debug:        # diagnostics: bit `strand` of the status word = this strand started, bit 16 + strand = it finished
            a("v_lshlrev_b32 v40, 2, v%d" % V_I)
            a("v_mov_b32 v41, 0x%x" % (1 << strand))
            a("global_atomic_or v40, v41, s[%d:%d]" % (S_STATUS, S_STATUS + 1))
            a("s_waitcnt vmcnt(0)")
        if prio:
            a("s_setprio 3")
        if any(st.call for st in steps):          # the interpreter body finds its tables through the kernel's argument block
            a("v_mov_b32 v40, s0")
            a("v_mov_b32 v41, s1")
            a("v_mov_b32 v42, 0")
            a("ds_write_b64 v42, v[40:41] offset:%d" % (park_off + 512))
            self.lg_issued += 1
        n = len(steps)

        def regs_of(k):
            p = k & 1
            return (FB.A_E, FB.B_E) if p == 0 else (FB.A_O, FB.B_O)

        load_seq = {}        # step -> (vm sequence number of its last table load | None, lgkm sequence number | None)

        def issue_loads(k):
            st = steps[k]
            ra, rb = regs_of(k)
            if st.heavy:
                ra, rb = FB.A_E, FB.B_E
            vm = lg = None
            for j, (which, kind, idx) in enumerate(st.loads):
                self.issue_load(ra if which == "A" else rb, kind, idx, 4 + 2 * j)
                if kind == K_LDS:
                    lg = self.lg_issued
                else:
                    vm = self.vm_issued
            load_seq[k] = (vm, lg)

        issued = set()
        for k, st in enumerate(steps):
            self.stats["steps"] += 1
            if st.barrier is not None:
                # LDS traffic of this wave is complete before any wave passes; global stores stay in flight (the waves of a
                # workgroup share the CU's L1, which keeps a store ahead of another wave's later load - DESIGN 4.1 (6))
                a("s_waitcnt lgkmcnt(0)")
                a("s_barrier")
                self.ir.append(("bar", st.barrier))
                continue
            if k not in issued:
                issue_loads(k)
                issued.add(k)
            # prefetch the next step's operands unless a barrier or a heavy body separates the two
            nxt = k + 1
            if nxt < n and steps[nxt].barrier is None and not st.heavy and not steps[nxt].heavy:
                issue_loads(nxt)
                issued.add(nxt)
            vm, lg = load_seq[k]
            if vm is not None or lg is not None:
                parts = []
                if vm is not None:
                    parts.append("vmcnt(%d)" % min(self.vm_issued - vm, VMCNT_MAX))
                if lg is not None:
                    parts.append("lgkmcnt(%d)" % min(self.lg_issued - lg, LGKM_MAX))
                a("s_waitcnt " + " ".join(parts))
                self.ir.append(("wait", vm, lg))
            ra, rb = regs_of(k)
            par = "eo"[k & 1]
            if st.heavy:
                ra, rb, par = FB.A_E, FB.B_E, "h"
            for p in st.pre:
                if p[0] == "prev":
                    self.mov_fe(ra if p[1] == "A" else rb, FB.D_REG)
                elif p[0] == "const":
                    self.lit_fe({"A": ra, "B": rb, "G": FB.G_REG}[p[1]], p[2])
                elif p[0] == "zero":
                    self.lit_fe(FB.G_REG, 0)
                elif p[0] == "zacc":
                    for j in range(self.acc_zero, p[1]):
                        a("v_mov_b32 v%d, 0" % (FB.ACC_REG + j))
                    self.ir.append(("zacc", p[1], min(self.acc_zero, p[1])))
                    self.acc_zero = max(self.acc_zero, p[1])
            if st.sarg is not None:
                a("s_mov_b32 s%d, 0x%x" % (FB.S_ARG, st.sarg & 0xFFFFFFFF))
                a("s_mov_b32 s%d, 0x%x" % (FB.S_ARG + 1, st.sarg >> 32))
                self.ir.append(("sarg", st.sarg))
            if st.coef is not None:
                for j, w in enumerate(st.coef):
                    a("s_mov_b32 s%d, 0x%x" % (FB.S_COEF + j, w))
                self.ir.append(("coef", tuple(st.coef)))
            if st.inline is not None:
                d = FB.D_REG
                if st.inline[0] == "copy":
                    self.mov_fe(d, ra)
                elif st.inline[0] == "stashg":
                    self.mov_fe(FB.G_REG, ra)
                elif st.inline[0] == "nop":
                    pass
                elif st.inline[0] == "bits":
                    kbit = st.inline[1]
                    for j in range(1, 8):
                        a("v_mov_b32 v%d, 0" % (d + j))
                    prev_slot = None
                    for slot, nxt, lo_only in st.inline[2]:
                        kbit += 1 if nxt else 0
                        if kbit < 256:
                            a("v_bfe_u32 v%d, v%d, %d, 1" % (d, ra + (kbit >> 5), kbit & 31))
                        else:
                            a("v_mov_b32 v%d, 0" % d)
                        if prev_slot is not None and slot == prev_slot + 1:
                            # (the bits of a Num2Bits are consecutive signals: the next slot is one stride further)
                            a("s_add_u32 s4, s4, s%d" % S_STRIDE)
                            a("s_addc_u32 s5, s5, 0")
                        else:
                            self.table_addr(slot, 4)
                        prev_slot = slot
                        a("global_store_dwordx4 v%d, v[%d:%d], s[4:5]" % (V_VLO, d, d + 3))
                        self.vm_issued += 1
                        if not lo_only:
                            a("global_store_dwordx4 v%d, v[%d:%d], s[4:5]" % (V_VHI, d + 4, d + 7))
                            self.vm_issued += 1
                        a("s_nop 1")                  # the store has read D before the next bit overwrites it
                        self.stats["stores"] += 1
                    self.ir.append(("bits", ra, st.inline[1], st.inline[2]))
                else:
                    kbit = st.inline[1]
                    if kbit < 256:
                        a("v_bfe_u32 v%d, v%d, %d, 1" % (d, ra + (kbit >> 5), kbit & 31))
                    else:
                        a("v_mov_b32 v%d, 0" % d)
                    for j in range(1, 8):
                        a("v_mov_b32 v%d, 0" % (d + j))
                    self.ir.append(("bit", ra, kbit))
            else:
                name = st.body if st.body in self.bodies else "%s_%s" % (st.body, par)
                b = self.bodies[name]
                if st.heavy:
                    # a heavy body keeps nothing but the status word (inv_h not even that): what the emitted code owns waits in
                    # LDS (the fused check's finding, the status word) or is re-derived from the lane number afterwards
                    a("v_lshrrev_b32 v%d, 2, v%d" % (V_L16B, V_L16))
                    a("ds_write_b32 v%d, v%d offset:%d" % (V_L16B, V_FB, park_off + 256))
                    self.lg_issued += 1
                    if name == "inv_h":
                        a("ds_write_b32 v%d, v%d offset:%d" % (V_L16B, V_ST, park_off))
                        self.lg_issued += 1
                    if st.call:
                        fn, slot0, seq = st.call
                        a("v_mov_b32 v42, 0")
                        a("ds_read_b64 v[40:41], v42 offset:%d" % (park_off + 512))
                        a("s_waitcnt vmcnt(0) lgkmcnt(0)")               # the window's arguments are stored; the address is here
                        a("v_readfirstlane_b32 s%d, v40" % (FB.S_COEF + 2))
                        a("v_readfirstlane_b32 s%d, v41" % (FB.S_COEF + 3))
                        a("s_mov_b32 s%d, 0x%x" % (FB.S_ARG, fn))
                        a("s_mov_b32 s%d, 0x%x" % (FB.S_ARG + 1, slot0))
                        a("s_mov_b32 s%d, 0x%x" % (FB.S_COEF, seq))
                        self.lg_issued += 1
                        self.ir.append(("callfn", fn, slot0, seq))
                    if b.scratch:
                        a("s_waitcnt vmcnt(0)")
                    self.call(name)
                    if b.scratch or st.call:
                        a("s_waitcnt vmcnt(0) lgkmcnt(0)")
                    self.rederive()
                    a("v_lshrrev_b32 v%d, 2, v%d" % (V_FB, V_L16))
                    if name == "inv_h":
                        a("ds_read_b32 v%d, v%d offset:%d" % (V_ST, V_FB, park_off))
                        self.lg_issued += 1
                    a("ds_read_b32 v%d, v%d offset:%d" % (V_FB, V_FB, park_off + 256))
                    a("s_waitcnt lgkmcnt(0)")
                    self.lg_issued += 1
                    self.ir.append(("heavy_done",))
                else:
                    self.call(name)
            for kind, idx in st.stores:
                self.issue_store(kind, idx)
            if any(kind != K_LDS for kind, _ in st.stores):
                a("s_nop 1")      # a 128-bit store must have read its data registers before anything writes D again
        # end of the strand: the first failed check of every instance reaches the status array
        if debug:
            a("s_waitcnt vmcnt(0)")
            a("v_lshlrev_b32 v40, 2, v%d" % V_I)
            a("v_mov_b32 v41, 0x%x" % (1 << (16 + strand)))
            a("global_atomic_or v40, v41, s[%d:%d]" % (S_STATUS, S_STATUS + 1))
            a("s_waitcnt vmcnt(0)")
            a("s_endpgm")
        self.call("publish")
        a("s_endpgm")
        return self.ir


def emit(tape, bodies=None, constraints=None, spool_path=None) -> FpJitProgram:
    """the emitted kernel of one schedule variant (strand schedule, kind 0); constraints = the circuit's R1CS rows
    (FlatCircuit.constraints) to fuse their check into the code (plan_checks); spool_path: write the text to this file as it
    is produced instead of keeping it (and the replay IR) in memory - for circuits of millions of rows"""
    if getattr(tape, "kind", 0) != 0:
        raise ValueError("only strand schedules have an emitted form")
    S = tape.n_strands
    assert S & (S - 1) == 0 and 1 <= S <= 16
    # (Round 6: schedules of several strands with run-time function calls, and functions with the native long_div, have an emitted
    # form too - the interpreter body `call_k`, compiled for the strand kernels' 128 VGPRs with a private segment for its spills
    # (fpjit_bodies.py), and `D_BITS` rows as one step of many stores.  Until then the interpreting kernel ran them: config 5.)
    if bodies is None:
        bodies = FB.build_bodies()
    em = _Emitter(tape, bodies, spool_path)
    all_steps = [expand_steps(tape, s) for s in range(S)]
    covered = plan_checks(tape, constraints, all_steps) if constraints else []
    # strands that carry >= 80 % of the heaviest strand's work run at raised priority (as cw_eval_kernel's prio_mask)
    cost = []
    for steps in all_steps:
        c = 0
        for st in steps:
            if st.body:
                nm = st.body if st.body in bodies else st.body + "_e"
                c += min(bodies[nm].n_instr, 2000)
        cost.append(c)
    heaviest = max(cost) if cost else 0
    n_lds = int(tape.n_lds)
    park = 0                                     # status words parked around inv_h: 256 bytes per strand, then the slots
    L = em.L
    L.append('.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n.text\n.globl %s\n.p2align 8\n.type %s,@function\n%s:\n'
             % (KERNEL_NAME, KERNEL_NAME, KERNEL_NAME))
    a = em.add
    # s[0:1] kernarg, s2 = workgroup id; v0 = work-item id.  Kernel arguments: V, status, Bp, batch, lanes, pad, FpParams
    a("s_load_dwordx4 s[4:7], s[0:1], 0x0")
    a("s_load_dwordx4 s[8:11], s[0:1], 0x10")
    off = 0x20
    s = FB.S_PARAMS
    left = FB.N_PARAM_SGPRS
    while left:
        w = 16 if left >= 16 else 4 if left >= 4 else 1
        a("s_load_dword%s s%s, s[0:1], 0x%x" % ({16: "x16", 4: "x4", 1: ""}[w], "[%d:%d]" % (s, s + w - 1) if w > 1 else "%d" % s, off))
        s += w
        off += 4 * w
        left -= w
    a("v_lshrrev_b32 v1, 6, v0")
    a("s_nop 1")                                          # a VGPR write needs a wait state before v_readfirstlane reads it
    a("v_readfirstlane_b32 s3, v1")
    a("s_waitcnt lgkmcnt(0)")
    a("s_mov_b64 s[%d:%d], s[4:5]" % (S_VBASE, S_VBASE + 1))
    a("s_mov_b64 s[%d:%d], s[6:7]" % (S_STATUS, S_STATUS + 1))
    a("s_lshl_b32 s%d, s8, 5" % S_STRIDE)                  # bytes per value slot = 2 * Bp * 16
    a("s_mov_b32 s%d, s9" % S_BATCH)
    a("s_mul_i32 s%d, s2, s10" % S_WGBASE)                 # first instance of the workgroup = wg * lanes
    # lanes beyond `lanes` are idle for the whole kernel (small batches of long schedules are spread over more workgroups)
    a("s_cmp_ge_u32 s10, 64")
    a("s_cbranch_scc1 .Lfull_wave")
    a("s_bfm_b64 exec, s10, 0")
    L.append(".Lfull_wave:\n")
    # strand of this wave: rotated by the workgroup index (the critical strand does not sit on the same SIMD everywhere)
    a("s_add_u32 s3, s3, s2")
    a("s_and_b32 s3, s3, %d" % (S - 1))
    em.strand = 0
    em.rederive()
    a("v_mov_b32 v%d, 0" % V_ST)
    a("v_mov_b32 v%d, -1" % V_FB)                        # no constraint found violated yet
    a("s_mov_b64 s[%d:%d], 0" % (FB.S_SEL, FB.S_SEL + 1))
    for k in range(8):
        a("v_mov_b32 v%d, 0" % (FB.D_REG + k))
    for s_ in range(1, S):
        a("s_cmp_eq_u32 s3, %d" % s_)
        a("s_cbranch_scc1 .Ljump_%d" % s_)
    if S > 1:
        a("s_branch fj_strand_0")
        for s_ in range(1, S):
            L.append(".Ljump_%d:\n" % s_)
            a("s_getpc_b64 s[4:5]")
            L.append(".Ljpc_%d:\n" % s_)
            a("s_add_u32 s4, s4, fj_strand_%d-.Ljpc_%d" % (s_, s_))
            a("s_addc_u32 s5, s5, 0")
            a("s_setpc_b64 s[4:5]")
    prog = FpJitProgram()
    prog.ir = []
    for s_ in range(S):
        prio = S > 1 and cost[s_] > 0 and cost[s_] >= 0.8 * heaviest
        prog.ir.append(em.strand_code(s_, all_steps[s_], prio, park + PARK_STRAND * s_))
    # the bodies this schedule calls
    scratch = 0
    for name in sorted(em.used_bodies):
        b = bodies[name]
        L.append(".p2align 6\nfj_body_%s:\n" % name)
        ret = "  s_setpc_b64 s[%d:%d]" % (S_RET, S_RET + 1)
        for t in b.text:
            L.extend(x + "\n" for x in ([ret] if t == FB.RET_MARK else FB.expand_long_branch(t)))
        if FB.RET_MARK not in b.text:
            L.append(ret + "\n")
        scratch = max(scratch, b.scratch_bytes)
    L.append(".Lend:\n.size %s, .Lend-%s\n" % (KERNEL_NAME, KERNEL_NAME))
    lds_bytes = PARK_BYTES + n_lds * LDS_SLOT
    n_vgpr, n_agpr = FB.N_VGPR, 0
    if "call_h" in em.used_bodies:                # the interpreter body was compiled for one wave per workgroup: up to 256 + 256
        assert S == 1, "call_h is the single-strand body"
        n_vgpr, n_agpr = 256, (bodies["call_h"].n_agpr + 7) // 8 * 8
    L.append(".rodata\n.p2align 6\n.amdhsa_kernel %s\n"
             "  .amdhsa_user_sgpr_kernarg_segment_ptr 1\n  .amdhsa_system_sgpr_workgroup_id_x 1\n  .amdhsa_system_vgpr_workitem_id 0\n"
             "  .amdhsa_next_free_vgpr %d\n  .amdhsa_accum_offset %d\n  .amdhsa_next_free_sgpr 102\n  .amdhsa_reserve_vcc 1\n"
             "  .amdhsa_group_segment_fixed_size %d\n  .amdhsa_private_segment_fixed_size %d\n%s  .amdhsa_kernarg_size %d\n"
             ".end_amdhsa_kernel\n" % (KERNEL_NAME, n_vgpr + n_agpr, n_vgpr, lds_bytes, scratch,
                                       "  .amdhsa_enable_private_segment 1\n" if scratch else "", KERNARG_BYTES))
    L.append(".amdgpu_metadata\n---\namdhsa.version: [1, 2]\namdhsa.kernels:\n  - .name: %s\n    .symbol: %s.kd\n"
             "    .kernarg_segment_size: %d\n    .group_segment_fixed_size: %d\n    .private_segment_fixed_size: %d\n"
             "    .kernarg_segment_align: 8\n    .wavefront_size: 64\n    .sgpr_count: 108\n    .vgpr_count: %d\n    .agpr_count: %d\n"
             "    .max_flat_workgroup_size: %d\n    .args:\n"
             "      - {.size: 8, .offset: 0, .value_kind: global_buffer, .address_space: global}\n"
             "      - {.size: 8, .offset: 8, .value_kind: global_buffer, .address_space: global}\n"
             "      - {.size: %d, .offset: 16, .value_kind: by_value}\n"
             "...\n.end_amdgpu_metadata\n" % (KERNEL_NAME, KERNEL_NAME, KERNARG_BYTES, lds_bytes, scratch, n_vgpr + n_agpr, n_agpr, 64 * S,
                                             KERNARG_BYTES - 16))
    prog.n_strands = S
    prog.asm = L.finish()
    prog.asm_path = spool_path if prog.asm is None else None
    prog.lds_bytes = lds_bytes
    prog.scratch_bytes = scratch
    prog.n_vgpr = n_vgpr + n_agpr
    prog.covered = covered
    prog.stats = dict(em.stats, bodies=len(em.used_bodies), n_lds=n_lds, constraints=len(covered), constraints_fused=sum(covered),
                      check_steps=sum(1 for steps in all_steps for st in steps if st.check),
                      same_register=getattr(plan_checks, "stats", {}).get("same_register", 0) if constraints else 0)
    return prog


def assemble(prog: FpJitProgram) -> bytes:
    from .bitjit import assemble as _asm, _llvm_bin
    if getattr(prog, "asm_path", None):           # spooled text: assemble the file where it lies
        import subprocess
        llvm, s = _llvm_bin(), prog.asm_path
        subprocess.run([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", s + ".o"],
                       check=True, capture_output=True)
        subprocess.run([os.path.join(llvm, "ld.lld"), "-shared", s + ".o", "-o", s + ".co"], check=True, capture_output=True)
        prog.code = open(s + ".co", "rb").read()
        for f in (s, s + ".o", s + ".co"):
            os.unlink(f)
        return prog.code
    prog.code = _asm(prog.asm)
    return prog.code
