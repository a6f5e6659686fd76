"""hip_elements lowering: flattened circuit -> batched signal-evaluation schedule ("tape").

This is the new back-end the north-star adds beside `c_elements`/`wasm_elements`
(code_producers/src/c_elements/mod.rs:6-39 is the producer it parallels): instead of emitting a
per-template C++/WASM program that interprets one input, it emits ONE schedule over global value slots
that the fixed HIP kernels (circom_amd/csrc/) evaluate for thousands of instances at once
(outer loop = schedule step, inner = instance/lane).

Representation policy ("N policy"): every value slot holds the CANONICAL residue in [0,q) as
8 x u32 limbs.  Rationale (vs. the reference's tagged short/long/Montgomery union, fr.hpp:17-21):
  * `.wtns` stores canonical values (main.cpp:326-332) -> no egress conversion pass,
  * bitwise/relational/shift operators are defined on canonical values,
  * add/sub are representation-agnostic,
  * a product with a compile-time constant is ONE raw Montgomery multiplication when the constant
    is pre-scaled by the device radix R' here (MMUL(x, c*R') = x*c), and a product of two run-time
    values is MMUL(MMUL(x,y), R'^2).

Passes:
  A expand      flat ops -> device ops over virtual temps (owns all Montgomery scaling)
  B alias       every `x <== y` that only copies a value becomes an extra destination of the row that
                produces y (at --O0 ~40% of all signals are such copies: component wiring).  The copy
                costs one more store (the algorithmic 32 B) instead of a load + a dependent store.
  C schedule    S parallel strands (waves of one workgroup working on the SAME 64 instances): rows are
                levelled by dependency depth, distributed over strands with producer affinity, and
                separated by workgroup barriers.  S = 1 keeps program order and needs no barrier.
  D slots       temp slot allocation by liveness (per row for S = 1, per barrier epoch for S > 1);
                a temp whose every use is forwarded through the register (kind PREV) gets no slot.
  D slots       cross-strand values get an LDS slot of the workgroup while they are live (hand-off in ~100
                cycles, and the barrier does not have to drain global stores); when the LDS pool is
                exhausted the hand-off goes through global memory and the barrier after the producing
                epoch is marked FULL (waits for vmcnt(0)).  Global temp slots by liveness; a temp whose
                every use is PREV/LDS gets none.
  E encode      rows are 4 x u32: w0 = op[0:8) | dk[8:11) | ak[11:14) | bk[14:17) | n_extra[17:29), then dst, a, b.
                operand kinds: 0 signal slot, 1 temp slot, 2 constant index, 3 PREV (the previous
                value-producing row of the same strand, forwarded in registers), 4 LDS slot;
                destination kinds: 0 signal, 1 temp, 2 none, 4 LDS.  The n_extra further destinations
                of a row (elided copies, LDS hand-off copy) are consecutive entries of the strand's
                extra-destination table; SELECT carries its third operand in an EXT row; BARRIER rows
                carry dst = 1 when they must also drain global stores.
  Invariant the kernel's one-row-ahead operand prefetch relies on: a row never reads FROM MEMORY a
  slot that the immediately preceding rows of its strand (back to and including the last
  value-producing row) write; such operands are always encoded as PREV.
"""
from __future__ import annotations

import os

import numpy as np

from .. import opcodes as O
from ..frontend.flatten import FlatCircuit

# device opcodes (csrc/cw_tape.h must match)
(D_COPY, D_ADD, D_SUB, D_NEG, D_MMUL, D_INV, D_IDIV, D_MOD, D_POW, D_SHL, D_SHR, D_BAND, D_BOR, D_BXOR,
 D_BNOT, D_LT, D_GT, D_LEQ, D_GEQ, D_EQ, D_NEQ, D_LAND, D_LOR, D_LNOT, D_SELECT, D_EXT, D_ASSERT_EQ,
 D_ASSERT_NZ, D_ALSO, D_BARRIER, D_MUL2, D_MADD, D_MULC, D_MADDC, D_LINSUM, D_BIT, D_DOTC, D_CALL, D_BITS) = range(39)
D_NAMES = ["copy", "add", "sub", "neg", "mmul", "inv", "idiv", "mod", "pow", "shl", "shr", "band", "bor",
           "bxor", "bnot", "lt", "gt", "leq", "geq", "eq", "neq", "land", "lor", "lnot", "select", "ext",
           "assert_eq", "assert_nz", "also", "barrier", "mul2", "madd", "mulc", "maddc", "linsum", "bit", "dotc", "call", "bits"]
# D_BITS  : consecutive bits of operand a starting at bit b (a raw number): one row for a whole Num2Bits instead of one D_BIT
#           row per bit.  No destination of its own; its extra-destination entries list, bit after bit, where each bit goes
#           (X_NEXT on an entry = "the next bit starts here": the bit's signal, then its elided copies).  Only in schedules
#           the interpreting kernel runs (circuits with run-time functions: pass A7 _fuse_bits); D_ALSO = a row that does
#           nothing, put between a D_BITS row and a following row of its strand that takes one of the bits as a (prefetched)
#           a / b operand.
# D_CALL  : run the register bytecode of a circom function (frontend/rtcode.py) per lane: field a = function id, operand b
#           = first of the function's registers, which are CONSECUTIVE pinned temp slots (arguments were stored there by
#           ordinary rows, results are read from there by the rows after the BARRIER that follows every D_CALL: the
#           interpreter's stores are not visible to an operand prefetched before the call).  Schedules with a D_CALL are
#           lowered for one strand only.
# function bytecode (device form): 4 x u32 per instruction: opcode (D_* for arithmetic, F_* below), dst, a, b; operands:
#           register number, or FN_CONST | index into the constant table; F_LDX / F_STX: b = index register | length << 16
F_JZ, F_JMP, F_LDX, F_STX, F_RET, F_DIV = 100, 101, 102, 103, 104, 105
FN_CONST = 1 << 31
# D_DOTC  : d = c0 + sum_i coef_i * x_i with ARBITRARY field coefficients: like D_LINSUM, but the second word of a
#           term indexes the limb-form constant table (coef_i * R' as 9 x 29-bit limbs); the kernel accumulates the
#           unreduced 29-bit-limb products of up to 4 terms and performs ONE Montgomery reduction for them
#           (Poseidon's MDS row = 3 products + 1 reduction instead of 3 full Montgomery multiplications + 2 adds).
# D_LINSUM: d = c0 + sum_i coef_i * x_i with small signed integer coefficients (|coef| < 2^63): field `a` = number of
#           terms, operand b = constant c0 (or none); the terms (operand, coefficient) are consecutive entries of
#           the strand's term table.  One row replaces the 2n-1 MULC/ADD/SUB rows `lin += in[j][k] * e2` loops trace into.
# D_BIT   : d = (a >> k) & 1, k = small constant in field b (a raw number, not a constant index)
# D_MULC : d = a * c on canonical values, c a compile-time constant.  Operand b = index of a constant PAIR:
#          [b] = c*R' (so that MMUL(a,[b]) = a*c), [b+1] = |val(c)| when c is a small signed integer.  Word-0
#          flags CS_POS / CS_NEG say so; the kernel then multiplies small run-time values directly
#          (the reference's short-int path, fr.cpp:416-439, chosen per wave at run time).
# D_MADDC: d = a * c + PREV, same operand convention.
# D_MUL2: d = a*b on canonical values  (= MMUL(MMUL(a,b), R'^2), the intermediate stays in registers)
# D_MADD: d = MMUL(a,b) + PREV          (multiply-accumulate of `lc += coeff * signal` chains)
# D_SELECT: latches the lane mask cond != 0 (no value); the following D_EXT row yields mask ? a : b

K_SIG, K_TMP, K_CONST, K_NONE = O.K_SIG, O.K_TMP, O.K_CONST, O.K_NONE
KD_NONE = 2      # destination kind "no store"
KO_PREV = 3      # operand kind "previous result of this strand"

_DIRECT = {O.COPY: D_COPY, O.ADD: D_ADD, O.SUB: D_SUB, O.NEG: D_NEG, O.IDIV: D_IDIV, O.MOD: D_MOD,
           O.POW: D_POW, O.SHL: D_SHL, O.SHR: D_SHR, O.BAND: D_BAND, O.BOR: D_BOR, O.BXOR: D_BXOR,
           O.BNOT: D_BNOT, O.LT: D_LT, O.GT: D_GT, O.LEQ: D_LEQ, O.GEQ: D_GEQ, O.EQ: D_EQ, O.NEQ: D_NEQ,
           O.LAND: D_LAND, O.LOR: D_LOR, O.LNOT: D_LNOT, O.ASSERT_EQ: D_ASSERT_EQ, O.ASSERT_NZ: D_ASSERT_NZ}
# (unit ~ 500 clocks of a lone wave, tools/profile_ops.sh on the ECDSA verifier: a bit row 3.2 K clk, mulc 4.9 K, mul2 5.2 K,
#  idiv / mod 15 K with Knuth D, inv 136 K)
_COST = {D_MMUL: 10.0, D_MUL2: 20.0, D_MADD: 11.0, D_MULC: 10.0, D_MADDC: 11.0, D_INV: 250.0, D_POW: 6000.0, D_IDIV: 30.0, D_MOD: 30.0}
_NO_VALUE = (D_ASSERT_EQ, D_ASSERT_NZ, D_SELECT)
# rows that can set the status word of an instance; the word carries the index of the flat operation (24 bits), and when
# several checks of one instance fail - on any strand, in any schedule order - the smallest index is reported: the check
# the reference's sequential program would have stopped at (assert_bucket.rs:75-77)
_FAIL_OPS = (D_ASSERT_EQ, D_ASSERT_NZ, D_IDIV, D_MOD, D_CALL)
SEQ_MAX = 0xFFFFFF


class Tape:
    """The lowered schedule + tables, ready to serialise (writers.py) or hand to the runtime."""

    def __init__(self):
        self.prime = ""
        self.q = 0
        self.n_signals = 0
        self.n_tslots = 0
        self.n_witness = 0
        self.rows = None            # (n,4) uint32, all strands concatenated
        self.stream_off = None      # uint32[n_strands+1] row offsets of each strand
        self.extras = None          # uint32[]: extra destinations, in stream/row order
        self.extra_off = None       # uint32[n_strands+1]: first extra-destination entry of each strand
        self.n_lds = 0              # LDS value slots the workgroup needs
        self.terms = None           # uint32[n,4]: D_LINSUM terms (kind|sign, index, |coef| lo, hi), stream/row order
        self.term_off = None        # uint32[n_strands+1]
        self.seqs = np.zeros(0, dtype=np.uint32)      # flat-operation index of every row that can fail, stream/row order
        self.seq_off = np.zeros(2, dtype=np.uint32)   # uint32[n_strands+1]
        self.lconsts = []           # constants of D_DOTC terms (coef * R' mod q), stored by the runtime as 29-bit limbs
        self.n_strands = 1
        self.consts = []            # raw residues (python ints)
        self.witness2signal = None  # uint32[n_witness]
        self.inputs = []            # (name, slot, size)
        self.main_input_start = 0
        self.n_main_inputs = 0
        self.n_pub_in = 0           # public inputs of main (`component main {public [...]}`); outputs are always public
        self.stats = {}
        self.rbits = 261            # Montgomery radix exponent of MMUL rows
        self.mont = False           # signals (value table) in Montgomery form x R' mod q (pass A6)
        self.kind = 0               # 0 = strand schedule (passes C/D), 1 = pipelined single-wave schedule (pipe.py)
        self.pipe = (0, 0, 0)       # kind 1: rows per batch, loads per batch, ring entries
        self.functions = []         # device bytecode of circom functions: (n_regs, uint32[n,4])
        self.log_strings = []       # string table of the log statements
        self.log_prog = []          # [(flat operation that ends the statement, [("s", string id) | ("v", j-th hidden signal)])]


def _dce(code, n_temps, nregs=()):
    op = code["op"]
    n = len(op)
    keep = np.zeros(n, dtype=bool)
    live = np.zeros(max(n_temps, 1), dtype=bool)
    dk, dv = code["dk"], code["dv"]
    cols = ((code["ak"], code["av"]), (code["bk"], code["bv"]), (code["ck"], code["cv"]))
    for i in range(n - 1, -1, -1):
        o = op[i]
        if o == O.CALL:             # the call and every register of its window (arguments are stored there)
            keep[i] = True
            live[code["bv"][i]:code["bv"][i] + nregs[code["av"][i]]] = True
            continue
        need = (o == O.ASSERT_EQ or o == O.ASSERT_NZ or dk[i] == K_SIG or (dk[i] == K_TMP and live[dv[i]]))
        if need:
            keep[i] = True
            for kk, vv in cols:
                if kk[i] == K_TMP:
                    live[vv[i]] = True
    return keep


class _Row:
    __slots__ = ("op", "dk", "dv", "ak", "av", "bk", "bv", "ck", "cv", "extra", "level", "strand", "flag", "coef", "terms", "seq", "multi")

    def __init__(self, op, dk, dv, ak, av, bk=K_NONE, bv=0, ck=K_NONE, cv=0):
        self.op, self.dk, self.dv = op, dk, dv
        self.ak, self.av, self.bk, self.bv, self.ck, self.cv = ak, av, bk, bv, ck, cv
        self.extra = None        # list of (kind, id) extra destinations
        self.flag = 0            # D_MULC/D_MADDC: 1 = constant is a small positive integer, 2 = small negative
        self.coef = None         # D_MULC: the plain constant (python int, canonical)
        self.terms = None        # D_LINSUM: list of [kind, id, signed coefficient]
        self.multi = None        # D_BITS: [(signal, [extra destinations] | None)] per bit
        self.level = 0
        self.strand = 0
        self.seq = 0             # index of the flat operation this row comes from (reported when the row fails a check)


def _proved_asserts(code, constants):
    """Range analysis for the commonest run-time check of bit-level circuits: `out * (out - 1) === 0` right after
    `out <-- (x >> k) & 1` (circomlib BinSum / Num2Bits / comparators).  `y & 1` is 0 or 1 for EVERY y, so the
    product is identically zero and the assert can never fire: it is dropped from the schedule (the R1CS check
    kernel still verifies the constraint itself).  Returns a boolean mask of ASSERT_EQ rows proved true."""
    op = code["op"]
    n = len(op)
    dk, dv = code["dk"], code["dv"]
    ak, av, bk, bv = code["ak"], code["av"], code["bk"], code["bv"]
    proved = np.zeros(n, dtype=bool)
    is_bit = {}          # (kind, id) -> True for values known to be 0/1
    prod = {}            # temp id -> row that defines it
    one = {i for i, c in enumerate(constants) if c == 1}
    zero = {i for i, c in enumerate(constants) if c == 0}

    def bit(k, v):
        if k == K_CONST:
            return v in one or v in zero
        return is_bit.get((int(k), int(v)), False)

    for i in range(n):
        o = op[i]
        if o == O.BAND and ((bk[i] == K_CONST and bv[i] in one) or (ak[i] == K_CONST and av[i] in one)):
            is_bit[(int(dk[i]), int(dv[i]))] = True
        elif o == O.COPY and bit(ak[i], av[i]):
            is_bit[(int(dk[i]), int(dv[i]))] = True
        elif o in (O.LT, O.GT, O.LEQ, O.GEQ, O.EQ, O.NEQ, O.LAND, O.LOR, O.LNOT):
            is_bit[(int(dk[i]), int(dv[i]))] = True
        if dk[i] == K_TMP:
            prod[int(dv[i])] = i
        if o == O.ASSERT_EQ:
            # ASSERT_EQ(t, 0) with t = MUL(x, SUB(x, 1)), x a bit
            for (tk, tv, zk, zv) in ((ak[i], av[i], bk[i], bv[i]), (bk[i], bv[i], ak[i], av[i])):
                if zk == K_CONST and zv in zero and tk == K_TMP and int(tv) in prod:
                    m = prod[int(tv)]
                    if op[m] != O.MUL:
                        continue
                    for (xk, xv, yk, yv) in ((ak[m], av[m], bk[m], bv[m]), (bk[m], bv[m], ak[m], av[m])):
                        if yk == K_TMP and int(yv) in prod and bit(xk, xv):
                            sb = prod[int(yv)]
                            if (op[sb] == O.SUB and (ak[sb], av[sb]) == (xk, xv) and bk[sb] == K_CONST
                                    and bv[sb] in one):
                                proved[i] = True
    return proved


def _expand(fc: FlatCircuit):
    """Pass A."""
    fp = fc.fp
    q = fp.q
    code = fc.code
    proved = _proved_asserts(code, fc.constants)
    if proved.any():
        code = dict(code)
        code["op"] = np.where(proved, np.uint8(O.RUN), code["op"])      # RUN rows are ignored by DCE and expansion
    keep = _dce(code, fc.n_temps, [f["n_regs"] for f in getattr(fc, "functions", ())])
    keep &= ~proved
    idx = np.nonzero(keep)[0]
    op = code["op"][idx].tolist()
    dk = code["dk"][idx].tolist(); dv = code["dv"][idx].tolist()
    ak = code["ak"][idx].tolist(); av = code["av"][idx].tolist()
    bk = code["bk"][idx].tolist(); bv = code["bv"][idx].tolist()
    ck = code["ck"][idx].tolist(); cv = code["cv"][idx].tolist()
    consts_in = fc.constants
    dconsts, dconst_id = [], {}
    plain = {}          # device constant id -> canonical value (only for constants used as plain operands)

    def cid(v):
        i = dconst_id.get(v)
        if i is None:
            i = len(dconsts)
            dconst_id[v] = i
            dconsts.append(v)
            plain[i] = v
        return i

    R, R2 = fp.Rdev, fp.Rdev2
    nxt = [fc.n_temps]
    pair_id = {}
    small_flag = {}

    def cpair(c):
        """constant pair for D_MULC/D_MADDC: scaled value, then the magnitude of the signed value if small"""
        c %= q
        i = pair_id.get(c)
        if i is None:
            mag, flag = 0, 0
            if c < (1 << 63):
                mag, flag = c, 1
            elif q - c < (1 << 63):
                mag, flag = q - c, 2
            i = len(dconsts)
            dconsts.append((c * R) % q)
            dconsts.append(mag)
            pair_id[c] = i
            small_flag[i] = flag
        return i, small_flag[i]


    def fresh():
        t = nxt[0]
        nxt[0] += 1
        return t

    def opnd(k, v):
        if k == K_CONST:
            return K_CONST, cid(consts_in[v] % q)
        return k, v

    rows = []
    n_pow2 = 0
    pow2_divisions = os.environ.get("CW_POW2_DIV", "1") != "0"
    row_seq, cur_seq = [], 0        # rows[k] comes from flat operation row_seq[k] (eval_flat's index of a failing check)
    flat_index = idx.tolist()
    for i in range(len(op)):
        o = op[i]
        while len(row_seq) < len(rows):
            row_seq.append(cur_seq)
        cur_seq = flat_index[i]
        if o == O.MUL:
            a_c, b_c = ak[i] == K_CONST, bk[i] == K_CONST
            if a_c or b_c:
                if a_c:
                    xk, xv, c = bk[i], bv[i], consts_in[av[i]]
                else:
                    xk, xv, c = ak[i], av[i], consts_in[bv[i]]
                ci, fl = cpair(c)
                r_ = _Row(D_MULC, dk[i], dv[i], xk, xv, K_CONST, ci)
                r_.flag = fl
                r_.coef = c % q
                rows.append(r_)
            else:
                rows.append(_Row(D_MUL2, dk[i], dv[i], ak[i], av[i], bk[i], bv[i]))
        elif o == O.DIV:
            # a / b = a * inv(b); inv(0) = 0 (generic/fr.cpp:2895-2912)
            t = fresh()
            kb, vb = opnd(bk[i], bv[i])
            rows.append(_Row(D_INV, K_TMP, t, kb, vb))
            if ak[i] == K_CONST:
                rows.append(_Row(D_MMUL, dk[i], dv[i], K_TMP, t, K_CONST, cid((consts_in[av[i]] * R) % q)))
            else:
                rows.append(_Row(D_MUL2, dk[i], dv[i], ak[i], av[i], K_TMP, t))
        elif o == O.CALL:
            r_ = _Row(D_CALL, KD_NONE, 0, K_NONE, av[i], K_TMP, bv[i])
            # the arguments are operands of the call for every dependency analysis (they are read from memory)
            r_.terms = [[K_TMP, bv[i] + k, 0] for k in range(fc.functions[av[i]]["n_args"])]
            rows.append(r_)
        elif (o == O.IDIV or o == O.MOD) and bk[i] == K_CONST and consts_in[bv[i]] % q > 0 \
                and (consts_in[bv[i]] % q) & ((consts_in[bv[i]] % q) - 1) == 0 and pow2_divisions:
            # x \ 2^k = x >> k and x % 2^k = x & (2^k - 1) on canonical values (Fr_idiv / Fr_mod divide the residues as integers,
            # generic/fr.cpp; a constant non-zero divisor cannot fail): a shift / a mask instead of a Knuth-D division - the carry
            # chains of circom-ecdsa's big-integer gadgets (`t % 2^n`, `t \ 2^n` per limb) are serial, 20 K clocks per division
            c_ = consts_in[bv[i]] % q
            ka, va = opnd(ak[i], av[i])
            if o == O.IDIV:
                rows.append(_Row(D_SHR, dk[i] if dk[i] != K_NONE else KD_NONE, dv[i], ka, va, K_CONST, cid(c_.bit_length() - 1)))
            else:
                rows.append(_Row(D_BAND, dk[i] if dk[i] != K_NONE else KD_NONE, dv[i], ka, va, K_CONST, cid(c_ - 1)))
            n_pow2 += 1
        elif o == O.SELECT:
            ka, va = opnd(ak[i], av[i])
            kb, vb = opnd(bk[i], bv[i])
            kc, vc = opnd(ck[i], cv[i])
            rows.append(_Row(D_SELECT, KD_NONE, 0, ka, va))
            rows.append(_Row(D_EXT, dk[i], dv[i], kb, vb, kc, vc))
        else:
            ka, va = opnd(ak[i], av[i])
            kb, vb = opnd(bk[i], bv[i]) if bk[i] != K_NONE else (K_NONE, 0)
            rows.append(_Row(_DIRECT[o], dk[i] if dk[i] != K_NONE else KD_NONE, dv[i], ka, va, kb, vb))
    while len(row_seq) < len(rows):
        row_seq.append(cur_seq)
    for r_, sq in zip(rows, row_seq):
        r_.seq = min(sq, SEQ_MAX)
    _expand.n_proved = int(proved.sum())
    _expand.n_pow2 = n_pow2
    return rows, dconsts, nxt[0], cid, plain


def _value_operands(r):
    """every (kind, id) a row reads, including D_LINSUM terms"""
    ops = [(r.ak, r.av), (r.bk, r.bv), (r.ck, r.cv)]
    if r.terms:
        ops.extend((t[0], t[1]) for t in r.terms)
    return ops


def _fuse_linear(rows, consts_plain, q, cid):
    """Pass A3: collapse trees of ADD / SUB / NEG / MULC-by-small-constant over single-use temporaries into one
    D_LINSUM row (the value is the same linear combination; field addition is associative and commutative).
    Also BAND(SHR(x, k), 1) -> D_BIT.  `consts_plain[i]` = canonical value of device constant i (None for scaled)."""
    uses = {}
    for r in rows:
        for k, v in _value_operands(r):
            if k == K_TMP:
                uses[v] = uses.get(v, 0) + 1
    prod = {}
    for idx, r in enumerate(rows):
        if r.dk == K_TMP:
            prod[r.dv] = idx
    absorbed = [False] * len(rows)
    repl = {}
    half = q >> 1
    # signals that are ONE BIT of another signal (the BAND(SHR(x, k), 1) pattern below, recognised ahead of the walk): a sum that
    # puts the bits of x back together is x & (2^n - 1), whatever the size of its coefficients (see the fusion branch)
    bit_of = {}
    for r in rows:
        if r.op == D_BAND and r.dk == K_SIG and r.bk == K_CONST and consts_plain.get(r.bv) == 1 and r.ak == K_TMP and uses.get(r.av) == 1:
            pi = prod.get(r.av)
            if pi is not None and rows[pi].op == D_SHR and rows[pi].bk == K_CONST and rows[pi].extra is None and rows[pi].ak == K_SIG:
                kk = consts_plain.get(rows[pi].bv)
                if kk is not None and kk < 256:
                    bit_of[r.dv] = (rows[pi].av, kk)
    n_bitsum = 0

    def sval(c):            # signed value of a canonical constant
        return c - q if c > half else c

    n_lin = n_bit = 0
    for idx in range(len(rows) - 1, -1, -1):
        r = rows[idx]
        if absorbed[idx]:
            continue
        if r.op == D_BAND and r.bk == K_CONST and consts_plain.get(r.bv) == 1 and r.ak == K_TMP and uses.get(r.av) == 1:
            pi = prod.get(r.av)
            if pi is not None and not absorbed[pi] and rows[pi].op == D_SHR and rows[pi].bk == K_CONST \
                    and rows[pi].extra is None:
                kk = consts_plain.get(rows[pi].bv)
                if kk is not None and kk < 256:
                    absorbed[pi] = True
                    nr = _Row(D_BIT, r.dk, r.dv, rows[pi].ak, rows[pi].av, K_NONE, kk)
                    nr.extra = r.extra
                    repl[idx] = nr
                    n_bit += 1
                    continue
        if r.op not in (D_ADD, D_SUB):
            continue
        terms, took, c0, ok = [], [], 0, True
        stack = [(r.bk, r.bv, -1 if r.op == D_SUB else 1), (r.ak, r.av, 1)]
        while stack and ok:
            k, v, cf = stack.pop()
            if k == K_CONST:
                cv_ = consts_plain.get(v)
                if cv_ is None:
                    ok = False
                else:
                    c0 += cf * sval(cv_)
                continue
            if k == K_TMP and uses.get(v) == 1:
                pi = prod.get(v)
                if pi is not None and not absorbed[pi] and rows[pi].extra is None:
                    pr = rows[pi]
                    if pr.op == D_ADD or pr.op == D_SUB:
                        took.append(pi)
                        stack.append((pr.bk, pr.bv, -cf if pr.op == D_SUB else cf))
                        stack.append((pr.ak, pr.av, cf))
                        continue
                    if pr.op == D_NEG:
                        took.append(pi)
                        stack.append((pr.ak, pr.av, -cf))
                        continue
                    if pr.op == D_MULC:
                        took.append(pi)
                        # exact integer product of signed representatives; reduced mod q when it leaves the small range
                        stack.append((pr.ak, pr.av, cf * sval(pr.coef)))
                        continue
            terms.append([k, v, cf])
        if not ok or len(took) < 3 or len(terms) > 4000:
            continue
        if c0 == 0 and 2 <= len(terms) < q.bit_length():
            # sum_i 2^i bit_i(x), i = 0 .. n-1 (the `lc1 += out[i] * e2` loop of Num2Bits): that integer is x & (2^n - 1) and stays
            # below q - ONE operand instead of n, each of which is its own 32-byte slot of the value table.  (For n > 63 the
            # coefficients do not fit a D_LINSUM term either: the chain used to stay 2 n rows.)
            src, seen = None, 0
            for k, v, cf in terms:
                b = bit_of.get(v) if k == K_SIG else None
                if b is None or (src is not None and b[0] != src) or cf != (1 << b[1]) or (seen >> b[1]) & 1:
                    seen = -1
                    break
                src = b[0]
                seen |= 1 << b[1]
            if seen == (1 << len(terms)) - 1:
                for pi in took:
                    absorbed[pi] = True
                nr = _Row(D_BAND, r.dk, r.dv, K_SIG, src, K_CONST, cid(seen))
                nr.extra = r.extra
                repl[idx] = nr
                n_bitsum += 1
                continue
        small = all(abs(t[2]) < (1 << 63) for t in terms)
        if not small:
            if len(terms) > 16:          # long sums with field-sized coefficients: leave to the MADDC chain
                continue
            for t in terms:
                t[2] = sval(t[2] % q)
        for pi in took:
            absorbed[pi] = True
        nr = _Row(D_LINSUM if small else D_DOTC, r.dk, r.dv, K_NONE, len(terms))
        if c0 % q:
            nr.bk, nr.bv = K_CONST, cid(c0 % q)
        nr.terms = terms
        nr.extra = r.extra
        repl[idx] = nr
        n_lin += 1
    out = []
    for idx, r in enumerate(rows):
        if absorbed[idx]:
            continue
        out.append(repl.get(idx, r))
    _fuse_linear.n_bitsum = n_bitsum
    return out, n_lin, n_bit


def _reassociate(rows, n_vtemps):
    """Pass A2: re-balance chains of field additions into trees.

    Field addition is associative and commutative, so `(((x0+x1)+x2)+...)+xn` (what `lin += in[j][k]*e2`
    loops of circomlib's BinSum/Num2Bits trace into) can be summed as a balanced tree: identical value, depth
    log2(n) instead of n.  Only intermediate sums that are single-use temporaries are touched.  The tree is
    emitted at the position of the chain's last row (all leaves are defined before it)."""
    uses = {}
    for r in rows:
        for k, v in _value_operands(r):
            if k == K_TMP:
                uses[v] = uses.get(v, 0) + 1
    prod = {}
    for idx, r in enumerate(rows):
        if r.dk == K_TMP:
            prod[r.dv] = idx
    absorbed = [False] * len(rows)
    trees = {}
    nxt = n_vtemps
    for idx in range(len(rows) - 1, -1, -1):
        r = rows[idx]
        if absorbed[idx] or r.op != D_ADD:
            continue
        leaves = []
        took = []
        stack = [(r.bk, r.bv), (r.ak, r.av)]       # DFS, left operand first
        while stack:
            k, v = stack.pop()
            if k == K_TMP and uses.get(v) == 1:
                pi = prod.get(v)
                if pi is not None and not absorbed[pi] and rows[pi].op == D_ADD and rows[pi].extra is None:
                    pr = rows[pi]
                    took.append(pi)
                    stack.append((pr.bk, pr.bv))
                    stack.append((pr.ak, pr.av))
                    continue
            leaves.append((k, v))
        if len(took) < 2:
            continue
        for pi in took:
            absorbed[pi] = True
        # balanced pairwise reduction
        new_rows = []
        cur = leaves
        while len(cur) > 2:
            nx = []
            for j in range(0, len(cur) - 1, 2):
                t = nxt
                nxt += 1
                new_rows.append(_Row(D_ADD, K_TMP, t, cur[j][0], cur[j][1], cur[j + 1][0], cur[j + 1][1]))
                nx.append((K_TMP, t))
            if len(cur) % 2:
                nx.append(cur[-1])
            cur = nx
        root = _Row(D_ADD, r.dk, r.dv, cur[0][0], cur[0][1], cur[1][0], cur[1][1])
        root.extra = r.extra
        new_rows.append(root)
        trees[idx] = new_rows
    out = []
    for idx, r in enumerate(rows):
        if absorbed[idx]:
            continue
        if idx in trees:
            out.extend(trees[idx])
        else:
            out.append(r)
    return out, nxt


INV_WINDOW = 2048    # independent inversions at most this many rows behind the previous member join its batch
INV_GROUP = 64       # members per batch


def _batch_inversions(rows, n_vtemps, cid):
    """Pass A4: Montgomery's trick.  k mutually independent inversions, close together in program order (the two
    denominators of a BabyAdd, the Z's of a projective ladder, the lambdas of parallel ladder segments, ...) become ONE
    inversion of their product plus 3(k-1) multiplications, arranged as a product tree (depth 2 log2 k, so that the strands
    can share the work):  up: p = p_left * p_right;  t_root = 1 / p_root;  down: t_left = t * p_right, t_right = t * p_left.
    Independence: inversions whose ARGUMENTS have the same inversion depth (number of INV rows on the longest path from
    the inputs) cannot feed one another.
    The reference's inverse maps 0 to 0 (mpz_invert fails, generic/fr.cpp:2895-2906), and a zero member would
    poison the product, so each member enters as e = d + [d == 0] and leaves as 1/e - [d == 0].
    Consumers of an early member that sit before the last member in program order are moved behind the batch
    (the rows are in SSA form, so any order that respects the data flow computes the same values)."""
    depth = {}
    inv_idx = []
    sel_depth = 0
    n_calls = 0                          # D_CALL rows seen so far: a batch never spans one (a call writes its results into
    calls_at = {}                        # its register window behind the row's back: rows that read them cannot be re-ordered
    for idx, r in enumerate(rows):       # around a deferred call by the data-flow rule below)
        if r.op == D_CALL:
            n_calls += 1
        calls_at[idx] = n_calls
        lv = 0
        for k, v in _value_operands(r):
            if k == K_SIG or k == K_TMP:
                lv = max(lv, depth.get((k, v), 0))
        if r.op == D_SELECT:             # the EXT row that follows also depends on the latched condition
            sel_depth = lv
        elif r.op == D_EXT:
            lv = max(lv, sel_depth)
        is_inv = r.op == D_INV and r.extra is None
        if is_inv:
            inv_idx.append((idx, lv))
        out_lv = lv + 1 if r.op == D_INV else lv
        if r.dk in (K_SIG, K_TMP):
            depth[(r.dk, r.dv)] = out_lv
        if r.extra:
            for x in r.extra:
                depth[x] = out_lv
    open_group = {}                      # inversion depth -> current group (list of row indices)
    groups = []
    for idx, lv in inv_idx:
        g = open_group.get(lv)
        if g is None or idx - g[-1] > INV_WINDOW or len(g) >= INV_GROUP or calls_at[g[-1]] != calls_at[idx]:
            g = []
            groups.append(g)
            open_group[lv] = g
        g.append(idx)
    member = {}
    for g in groups:
        if len(g) >= 2:
            for idx in g:
                member[idx] = g
    if not member:
        return rows, n_vtemps, 0
    nxt = [n_vtemps]

    def tmp():
        t = nxt[0]
        nxt[0] += 1
        return (K_TMP, t)

    zero = (K_CONST, cid(0))
    out = []
    pending = set()                      # values whose producer has not been emitted yet
    deferred = []
    arrived = {}

    def emit_batch(g):
        mem = [rows[i] for i in g]
        z, e = [], []
        for r in mem:
            zi, ei = tmp(), tmp()
            out.append(_Row(D_EQ, zi[0], zi[1], r.ak, r.av, zero[0], zero[1]))
            out.append(_Row(D_ADD, ei[0], ei[1], r.ak, r.av, zi[0], zi[1]))
            z.append(zi)
            e.append(ei)
        # product tree, bottom up (an odd node is carried to the next level unchanged)
        levels = [e]
        while len(levels[-1]) > 1:
            cur, up = levels[-1], []
            for i in range(0, len(cur) - 1, 2):
                pi = tmp()
                out.append(_Row(D_MUL2, pi[0], pi[1], cur[i][0], cur[i][1], cur[i + 1][0], cur[i + 1][1]))
                up.append(pi)
            if len(cur) & 1:
                up.append(cur[-1])
            levels.append(up)
        t = tmp()
        out.append(_Row(D_INV, t[0], t[1], levels[-1][0][0], levels[-1][0][1]))
        inv = [t]
        for lv in range(len(levels) - 2, -1, -1):      # top down: the inverse of a node times its sibling's product
            cur, down = levels[lv], []
            for i in range(0, len(cur) - 1, 2):
                ti = inv[i // 2]
                tl, tr = tmp(), tmp()
                out.append(_Row(D_MUL2, tl[0], tl[1], ti[0], ti[1], cur[i + 1][0], cur[i + 1][1]))
                out.append(_Row(D_MUL2, tr[0], tr[1], ti[0], ti[1], cur[i][0], cur[i][1]))
                down.extend((tl, tr))
            if len(cur) & 1:
                down.append(inv[len(cur) // 2])
            inv = down
        for r, ri, zi in zip(mem, inv, z):
            out.append(_Row(D_SUB, r.dk, r.dv, ri[0], ri[1], zi[0], zi[1]))

    def dsts(unit):
        for r in unit:
            if r.dk in (K_SIG, K_TMP):
                yield (r.dk, r.dv)
            if r.extra:
                for x in r.extra:
                    yield x

    def process(unit, idx):
        blocked = any((k, v) in pending for r in unit for k, v in _value_operands(r))
        if blocked:
            deferred.append((unit, idx))
            pending.update(dsts(unit))
            return
        g = member.get(idx) if len(unit) == 1 and unit[0].op == D_INV else None
        if g is None:
            out.extend(unit)
            return
        arrived[id(g)] = arrived.get(id(g), 0) + 1
        pending.update(dsts(unit))
        if arrived[id(g)] < len(g):
            return
        emit_batch(g)
        for i in g:
            pending.discard((rows[i].dk, rows[i].dv))
        q = list(deferred)
        del deferred[:]
        for u, ui in q:
            for d in dsts(u):
                pending.discard(d)
        for u, ui in q:
            process(u, ui)

    i = 0
    while i < len(rows):
        if rows[i].op == D_SELECT:
            process((rows[i], rows[i + 1]), i)
            i += 2
        else:
            process((rows[i],), i)
            i += 1
    assert not deferred and not pending
    return out, nxt[0], sum(1 for g in groups if len(g) >= 2)


_INT_OPS = (D_IDIV, D_MOD, D_POW, D_SHL, D_SHR, D_BAND, D_BOR, D_BXOR, D_BNOT, D_LT, D_GT, D_LEQ, D_GEQ)


def _to_montgomery(rows, plain, cid, q, R, n_vtemps):
    """Pass A6 (arithmetic circuits: compiler.choose_mont): keep every SIGNAL and most temporaries in Montgomery form
    x~ = x R' mod q, so that a product of two run-time values is ONE Montgomery product (mmul(a~, b~) = ab R') instead of
    the two a canonical product needs (mmul(mmul(a, b), R'^2)).  Linear operators, comparisons with zero, equality and
    selection are the same on both forms; constants are re-scaled here; D_MMUL / D_MULC / D_DOTC already multiply by
    c R', which maps a~ to (ac)~.  Operators defined on canonical integers (bit extraction, shifts, bitwise, ordering, integer
    division, powers) get their operands converted (mmul(x~, 1) = x, cached per value) and produce integers; an integer
    that becomes a signal, or meets a Montgomery-form value in an addition or product, is converted back (mmul(x, R'^2)).
    The inverse of a~ is a^-1 R'^-1: one more product by R'^3.
    The runtime converts at its boundary: ingest multiplies the inputs by R'^2, egress by 1, the R1CS check compares
    mmul(A~, B~) with C~ (csrc/cw_kernels.hip).  Returns (rows, n_vtemps, number of conversion rows)."""
    M, I = 'M', 'I'
    R2, R3 = R * R % q, R * R * R % q
    dom = {}
    as_i, as_m = {}, {}
    out = []
    nxt = [n_vtemps]
    n_conv = [0]
    one, r2c, r3c = cid(1), cid(R2), cid(R3)

    def fresh():
        t = nxt[0]
        nxt[0] += 1
        return (K_TMP, t)

    def dom_of(k, v):
        return dom.get((k, v), M)           # main inputs and the constant-one signal: converted by the ingest kernel

    def conv(k, v, want):
        """operand (k, v) in the wanted domain"""
        if k == K_NONE:
            return k, v
        if k == K_CONST:
            return (K_CONST, cid(plain[v] * R % q)) if want == M else (k, v)
        if dom_of(k, v) == want:
            return k, v
        cache = as_i if want == I else as_m
        got = cache.get((k, v))
        if got is None:
            got = fresh()
            out.append(_Row(D_MMUL, got[0], got[1], k, v, K_CONST, one if want == I else r2c))
            dom[got] = want
            cache[(k, v)] = got
            n_conv[0] += 1
        return got

    def finish(r, d):
        """r computes a value of domain d into (r.dk, r.dv); signals hold the Montgomery form"""
        if r.dk == K_SIG and d == I:
            t = fresh()
            sig = (r.dk, r.dv)
            r.dk, r.dv = t
            out.append(r)
            dom[t] = I
            out.append(_Row(D_MMUL, sig[0], sig[1], t[0], t[1], K_CONST, r2c))
            dom[sig] = M
            as_i[sig] = t
            n_conv[0] += 1
            return
        out.append(r)
        if r.dk in (K_SIG, K_TMP):
            dom[(r.dk, r.dv)] = d

    def common(ops):
        ds = {dom_of(k, v) for k, v in ops if k in (K_SIG, K_TMP)}
        return I if ds == {I} else M

    for r in rows:
        op = r.op
        if op in (D_LINSUM, D_DOTC):
            d = common([(t[0], t[1]) for t in r.terms])
            if op == D_DOTC or d == M:
                d = M
                if op == D_LINSUM:
                    r.op = D_DOTC
                    for t in r.terms:
                        t[2] = t[2] % q if t[2] >= 0 else -((-t[2]) % q)
            for t in r.terms:
                t[0], t[1] = conv(t[0], t[1], d)
            if r.bk == K_CONST:
                r.bk, r.bv = conv(r.bk, r.bv, d)
            finish(r, d)
        elif op == D_BIT:
            r.ak, r.av = conv(r.ak, r.av, I)
            finish(r, I)
        elif op in _INT_OPS:
            r.ak, r.av = conv(r.ak, r.av, I)
            r.bk, r.bv = conv(r.bk, r.bv, I)
            finish(r, I)
        elif op in (D_EQ, D_NEQ, D_ASSERT_EQ):
            d = common([(r.ak, r.av), (r.bk, r.bv)])
            r.ak, r.av = conv(r.ak, r.av, d)
            r.bk, r.bv = conv(r.bk, r.bv, d)
            finish(r, I)
        elif op in (D_LAND, D_LOR, D_LNOT, D_SELECT, D_ASSERT_NZ):
            finish(r, I)                          # tests against zero: the same on both forms
        elif op in (D_ADD, D_SUB, D_NEG, D_COPY, D_EXT):
            d = common([(r.ak, r.av), (r.bk, r.bv)])
            r.ak, r.av = conv(r.ak, r.av, d)
            r.bk, r.bv = conv(r.bk, r.bv, d)
            finish(r, d)
        elif op == D_MUL2:
            d = common([(r.ak, r.av), (r.bk, r.bv)])
            if d == M:
                r.op = D_MMUL
                r.ak, r.av = conv(r.ak, r.av, M)
                r.bk, r.bv = conv(r.bk, r.bv, M)
            finish(r, d)
        elif op == D_MMUL or op == D_MULC:        # second operand = c R' (pre-scaled): a~ -> (ac)~, a -> ac
            d = dom_of(r.ak, r.av)
            if d == M:
                r.flag = 0                        # the small-constant shortcut multiplies canonical values
            finish(r, d)
        elif op == D_INV:
            if dom_of(r.ak, r.av) == I:
                finish(r, I)
            else:
                t = fresh()
                dst = (r.dk, r.dv)
                r.dk, r.dv = t
                out.append(r)
                dom[t] = M
                nr = _Row(D_MMUL, dst[0], dst[1], t[0], t[1], K_CONST, r3c)
                out.append(nr)
                dom[dst] = M
                n_conv[0] += 1
        else:
            raise ValueError("operator %d has no Montgomery-domain rule" % op)
    return out, nxt[0], n_conv[0]


LINSUM_SPLIT_MIN = 24     # D_LINSUM rows with at least this many terms are split across strands (S > 1)


def _split_linsums(rows, n_vtemps, n_strands):
    """Pass A5 (multi-strand schedules): a long D_LINSUM (the 32-bit adders of SHA-256 are sums of 64-160 bit
    terms) is one row of one strand whose operands arrive four at a time, i.e. tens of memory latencies in
    series on the critical path.  It becomes `n` partial sums of consecutive terms, which the scheduler spreads
    over the strands, plus one short sum of the partials: same linear combination, same value."""
    out = []
    nxt = n_vtemps
    n_split = 0
    for r in rows:
        if r.op != D_LINSUM or len(r.terms) < LINSUM_SPLIT_MIN:
            out.append(r)
            continue
        parts = min(n_strands, max(2, len(r.terms) // 8))
        per = -(-len(r.terms) // parts)
        per = (per + 3) // 4 * 4                       # operands are fetched four at a time
        partial = []
        for k in range(0, len(r.terms), per):
            chunk = r.terms[k:k + per]
            pr = _Row(D_LINSUM, K_TMP, nxt, K_NONE, len(chunk))
            pr.terms = chunk
            out.append(pr)
            partial.append([K_TMP, nxt, 1])
            nxt += 1
        fr = _Row(D_LINSUM, r.dk, r.dv, K_NONE, len(partial), r.bk, r.bv)
        fr.terms = partial
        fr.extra = r.extra
        out.append(fr)
        n_split += 1
    return out, nxt, n_split


def _fuse_madd(rows):
    """Pass B2 (single strand only): [x = value] [t = MMUL(a,b)] [d = x + t]  ->  [x] [d = MADD(a,b)] where the
    addend is implicitly PREV (= x).  `t` must be a single-use temp and `x` must be produced by the row right
    before the MMUL (so that it is still in the forwarding register)."""
    uses = {}
    for r in rows:
        for k, v in _value_operands(r):
            if k == K_TMP:
                uses[v] = uses.get(v, 0) + 1
    out = []
    i, n, fused = 0, len(rows), 0
    while i < n:
        r = rows[i]
        if (r.op in (D_MMUL, D_MULC) and r.dk == K_TMP and uses.get(r.dv) == 1 and r.extra is None and i + 1 < n and out):
            add, x = rows[i + 1], out[-1]
            if add.op == D_ADD and x.op not in _NO_VALUE and x.dk in (K_SIG, K_TMP):
                t_op, x_op = (K_TMP, r.dv), (x.dk, x.dv)
                ops = ((add.ak, add.av), (add.bk, add.bv))
                # x must not itself feed the product (it is only available as PREV, which MADD uses as addend)
                feeds = (r.ak, r.av) == x_op or (r.bk, r.bv) == x_op
                if not feeds and (ops == (x_op, t_op) or ops == (t_op, x_op)):
                    m = _Row(D_MADD if r.op == D_MMUL else D_MADDC, add.dk, add.dv, r.ak, r.av, r.bk, r.bv)  # addend = PREV
                    m.flag = r.flag
                    m.extra = add.extra
                    out.append(m)
                    fused += 1
                    i += 2
                    continue
        out.append(r)
        i += 1
    return out, fused


def _alias(rows, n_signals):
    """Pass B.  Value ids: signal s -> s, virtual temp t -> n_signals + t."""
    def vid(k, v):
        return v if k == K_SIG else n_signals + v

    producer = {}
    root = {}

    def find(x):
        while x in root:
            x = root[x]
        return x

    out = []
    n_elided = 0
    for r in rows:
        # canonicalise operands to the root of their alias class
        for kk, vv in (("ak", "av"), ("bk", "bv"), ("ck", "cv")):
            k = getattr(r, kk)
            if k == K_SIG or k == K_TMP:
                x = find(vid(k, getattr(r, vv)))
                if x < n_signals:
                    setattr(r, kk, K_SIG); setattr(r, vv, x)
                else:
                    setattr(r, kk, K_TMP); setattr(r, vv, x - n_signals)
        if r.terms:
            for t in r.terms:
                if t[0] == K_SIG or t[0] == K_TMP:
                    x = find(vid(t[0], t[1]))
                    if x < n_signals:
                        t[0], t[1] = K_SIG, x
                    else:
                        t[0], t[1] = K_TMP, x - n_signals
        if r.op == D_COPY and r.dk == K_SIG and r.ak in (K_SIG, K_TMP):
            src = vid(r.ak, r.av)
            p = producer.get(src)
            if p is not None and (p.extra is None or len(p.extra) < EXTRA_CAP):
                if p.extra is None:
                    p.extra = []
                p.extra.append((K_SIG, r.dv))
                root[r.dv] = src
                n_elided += 1
                continue
            if p is not None:
                # the producer's extra-destination field is full (a value wired into thousands of places): this copy
                # stays a row of its own and takes over as the carrier of the following copies of the same value
                producer[src] = r
                root[r.dv] = src
                out.append(r)
                continue
        if r.dk in (K_SIG, K_TMP):
            producer[vid(r.dk, r.dv)] = r
        out.append(r)
    return out, n_elided


MAX_BITS_ROW = 128


def _defs(r):
    """(kind, id) of every value a row defines (D_BITS: one per bit)"""
    if r.multi:
        return [(K_SIG, dv) for dv, _ in r.multi]
    return [(r.dk, r.dv)] if r.dk in (K_SIG, K_TMP) else []


def _fuse_bits(rows):
    """Pass A7 (schedules of the interpreting kernel): runs of D_BIT rows that take consecutive bits of the same operand into
    signals - what Num2Bits traces into - become one D_BITS row.  A 64-bit range check is then 1 row with 64 stores instead of
    64 rows that each fetch the operand again (the ECDSA verifier: 2.3 M of its 3.3 M rows are such bit rows)."""
    out, n_fused = [], 0
    i, n = 0, len(rows)
    while i < n:
        r = rows[i]
        if r.op == D_BIT and r.dk == K_SIG and r.ak in (K_SIG, K_TMP):
            j = i + 1
            ex = len(r.extra or ())
            while j < n and j - i < MAX_BITS_ROW:
                t = rows[j]
                if not (t.op == D_BIT and t.dk == K_SIG and t.ak == r.ak and t.av == r.av and t.bv == r.bv + (j - i)):
                    break
                if ex + len(t.extra or ()) + (j - i + 1) > EXTRA_CAP:
                    break
                ex += len(t.extra or ())
                j += 1
            if j - i >= 2:
                nr = _Row(D_BITS, KD_NONE, 0, r.ak, r.av, K_NONE, r.bv)
                nr.multi = [(t.dv, list(t.extra) if t.extra else None) for t in rows[i:j]]
                nr.seq = r.seq
                out.append(nr)
                n_fused += j - i
                i = j
                continue
        out.append(r)
        i += 1
    return out, n_fused


_COST_INTERP = {D_MULC: 8.0, D_MADDC: 8.0, D_MUL2: 10.0, D_MMUL: 7.5, D_MADD: 8.0, D_IDIV: 23.0, D_MOD: 23.0, D_INV: 180.0, D_POW: 6000.0}


def _schedule(rows, n_signals, n_strands, functions=(), interp_costs=False):
    """Pass C.  Returns (streams, number of barriers, ordinals of the barriers that must be FULL); each stream is a list whose
    items are _Row or the string 'B'.
    A D_CALL (a circom function with run-time control flow) reads its arguments from and writes its results to its register
    window in the VALUE TABLE: with several strands it is a heavy unit of its level on one strand, and the barriers in front of
    and behind that level drain global stores (FULL), so that arguments stored by any strand are in memory when the call
    starts and its results are when the next level reads them."""
    if n_strands <= 1:
        st = []
        nb = 0
        for r in rows:
            r.strand = 0
            st.append(r)
            if r.op == D_CALL:      # operands of the next row must not be prefetched before the call has run
                st.append("B")
                nb += 1
        return [st], nb, set()

    def vid(k, v):
        return v if k == K_SIG else n_signals + v

    prod_level = {}
    prod_strand = {}
    levels = []
    # SELECT latches a per-strand lane mask consumed by the EXT row that follows: schedule the pair as a unit
    units = []
    i = 0
    while i < len(rows):
        if rows[i].op == D_SELECT:
            units.append((rows[i], rows[i + 1]))
            i += 2
        else:
            units.append((rows[i],))
            i += 1
    for unit in units:
        lv = 0
        for r in unit:
            for k, v in _value_operands(r):
                if k == K_SIG or k == K_TMP:
                    pl = prod_level.get(vid(k, v))
                    if pl is not None and pl + 1 > lv:
                        lv = pl + 1
        for r in unit:
            r.level = lv
            for dk_, dv_ in _defs(r):
                prod_level[vid(dk_, dv_)] = lv
            if r.op == D_CALL:      # every register of the window is the call's from here on (results, scratch)
                for k in range(functions[r.av]["n_regs"]):
                    prod_level[vid(K_TMP, r.bv + k)] = lv
        while len(levels) <= lv:
            levels.append([])
        levels[lv].append(unit)
    streams = [[] for _ in range(n_strands)]

    def ucost_interp(unit):
        """schedules the interpreting kernel runs (circuits with run-time functions): MEASURED clocks per row of the 16-strand
        ECDSA verifier (tools/profile_ops.sh, profiles/r05c_ecdsa_prof.log; unit = 900 clk, four waves per SIMD): every row costs
        ~5 K clk whatever it computes, a term of a sum is a dependent table read, a call is a binary-GCD inverse or a long division"""
        c = 0.0
        for r in unit:
            if r.op == D_CALL:
                nat = functions[r.av].get("native")
                c += CALL_COST if not nat else 130.0 if nat[0] == "long_div" else 170.0
            elif r.op == D_BITS:
                c += 6.0 + BITS_ENTRY_COST * (len(r.multi) + sum(len(x or ()) for _, x in r.multi))
            elif r.op == D_LINSUM:
                c += 6.0 + 4.8 * len(r.terms)
            elif r.op == D_DOTC:
                c += 6.0 + 6.0 * len(r.terms)
            else:
                c += _COST_INTERP.get(r.op, 6.0)
            if r.extra:
                c += 0.5 * len(r.extra)
        return c

    def ucost(unit):      # ~ VALU instructions / 32, plus a fixed part per row (operand fetch + dispatch latency)
        if interp_costs:
            return ucost_interp(unit)
        c = 0.0
        for r in unit:
            c += ROW_OVERHEAD
            if r.op == D_CALL:
                nat = functions[r.av].get("native")
                c += CALL_COST if not nat else CALL_COST_LONG_DIV if nat[0] == "long_div" else CALL_COST_NATIVE
            elif r.op == D_BITS:
                # (per entry: a scalar table read shared by four, the bit, two 16-byte stores that nothing waits for)
                c += BITS_ENTRY_COST * (len(r.multi) + sum(len(x or ()) for _, x in r.multi))
            elif r.op == D_DOTC:
                c += 5.0 + 4.5 * len(r.terms)
            elif r.op == D_LINSUM:
                c += 3.0 + 1.2 * len(r.terms)
            else:
                c += _COST.get(r.op, 2.0)
            if r.extra:
                c += EXTRA_COST * len(r.extra)
        return c

    # A barrier is only needed in front of a level that reads, from ANOTHER strand, a value produced since the last
    # barrier; levels whose cross-strand inputs are all older run on without synchronising (dependent rows of one
    # strand simply follow each other in its stream).
    fresh = {}          # value id -> producing strand, for values produced since the last barrier
    n_barriers = 0
    forced_full = set()
    after_call = False
    for lv, lunits in enumerate(levels):
        has_call = any(u[0].op == D_CALL for u in lunits)
        total = sum(ucost(u) for u in lunits)
        cap = total / n_strands * (AFFINITY_SLACK_INTERP if interp_costs else AFFINITY_SLACK) + 4.0
        load = [0.0] * n_strands
        placed = []
        if interp_costs:
            # Interpreted rows cost ~5 K clocks each whatever they compute and hand everything over through memory or LDS anyway
            # (measured on the ECDSA verifier: 47 % of the strands' time was waiting at barriers, 70 % of a level the spread between
            # the first and the last strand to arrive): balance first - longest unit first onto the least loaded strand, the
            # producer of its operands only as the tie-break - and keep program order inside every strand.
            costs = [ucost(u) for u in lunits]
            where = [0] * len(lunits)
            for ui in sorted(range(len(lunits)), key=lambda i_: -costs[i_]):
                unit = lunits[ui]
                lo = min(load)
                best, best_votes = -1, -1
                votes = {}
                for r in unit:
                    for k, v in _value_operands(r):
                        if k == K_SIG or k == K_TMP:
                            ps = prod_strand.get(vid(k, v))
                            if ps is not None:
                                votes[ps] = votes.get(ps, 0) + 1
                for s_ in range(n_strands):
                    if load[s_] <= lo + 1.0 and votes.get(s_, 0) > best_votes:
                        best, best_votes = s_, votes.get(s_, 0)
                where[ui] = best
                load[best] += costs[ui]
            placed = [(u, where[i_]) for i_, u in enumerate(lunits)]
            lunits = ()
        for unit in lunits:
            cost = ucost(unit)
            # affinity: the strand that produced most of the unit's operands (most recent level first) keeps the
            # data flow inside one wave (register/own-store forwarding instead of a cross-strand hand-off)
            votes = {}
            for r in unit:
                for k, v in _value_operands(r):
                    if k == K_SIG or k == K_TMP:
                        x = vid(k, v)
                        ps = prod_strand.get(x)
                        if ps is not None:
                            votes[ps] = votes.get(ps, 0) + (4 if prod_level.get(x) == lv - 1 else 1)
            pref = None
            for ps, _ in sorted(votes.items(), key=lambda kv: -kv[1]):
                if load[ps] + cost <= cap:
                    pref = ps
                    break
            if pref is None:
                pref = min(range(n_strands), key=load.__getitem__)
            load[pref] += cost
            placed.append((unit, pref))
        need = False
        for unit, pref in placed:
            for r in unit:
                for k, v in _value_operands(r):
                    if (k == K_SIG or k == K_TMP) and fresh.get(vid(k, v), pref) != pref:
                        need = True
        if need or has_call or after_call:
            if has_call or after_call:
                forced_full.add(n_barriers)
            for st in streams:
                st.append("B")
            fresh.clear()
            n_barriers += 1
        after_call = has_call
        for unit, pref in placed:
            for r in unit:
                r.strand = pref
                streams[pref].append(r)
                for dk_, dv_ in _defs(r):
                    prod_strand[vid(dk_, dv_)] = pref
                    fresh[vid(dk_, dv_)] = pref
    return streams, n_barriers, forced_full


BITS_ENTRY_COST = float(os.environ.get("CW_BITS_ENTRY_COST", "0.3"))      # one destination of a D_BITS row (a 16th of a table read + ~20 instructions)
CALL_COST = float(os.environ.get("CW_CALL_COST", "20000"))               # an interpreted function (long_div: ~10^5 instructions)
CALL_COST_NATIVE = float(os.environ.get("CW_CALL_COST_NATIVE", "300"))   # a native routine (one binary-GCD inverse + products: ~150 K clk)
CALL_COST_LONG_DIV = float(os.environ.get("CW_CALL_COST_LONG_DIV", "40"))  # native long_div (a dozen Knuth digits)
ROW_OVERHEAD = float(os.environ.get("CW_ROW_OVERHEAD", "4"))   # scheduler cost units charged to every row
EXTRA_COST = float(os.environ.get("CW_EXTRA_COST", "2"))     # ... and to every extra destination (two 1-KiB stores)
FULL_PERIOD = 8        # every FULL_PERIOD-th barrier also drains global stores
AFFINITY_SLACK = float(os.environ.get("CW_AFFINITY_SLACK", "1.25"))
AFFINITY_SLACK_INTERP = float(os.environ.get("CW_AFFINITY_SLACK_INTERP", "1.0"))   # (interpreted rows: balance first, every row costs the same)  # a strand may take this much more than the average load of a level to keep data local


def lds_slots_for(n_strands: int) -> int:
    """LDS value slots (64 lanes x 32 B = 2 KiB each) a workgroup of `n_strands` waves may use: 144 KiB per CU
    shared by the 16/n_strands workgroups that fit a CU."""
    return 0 if n_strands <= 1 else (72 * n_strands) // 16


# row word 0 layout:  op[0:8) | dk[8:11) | ak[11:14) | bk[14:17) | n_extra[17:29) | const-small flags[29:31)
SH_DK, SH_AK, SH_BK, SH_NX, SH_FLAG = 8, 11, 14, 17, 29
MAX_EXTRA = 4095
EXTRA_CAP = 4000       # copies folded into one row (leaves room for the LDS hand-off entry; the field holds 4095)
K_LDS = 4             # operand / destination kind: LDS slot of the workgroup (cross-strand hand-off)
X_TMP, X_LDS = 1 << 31, 1 << 30       # flags of an entry of the extra-destination table
X_NEXT = 1 << 29                      # ... of a D_BITS row: this entry belongs to the next bit


_FN_ALU = {O.COPY: D_COPY, O.ADD: D_ADD, O.SUB: D_SUB, O.NEG: D_NEG, O.MUL: D_MUL2, O.DIV: F_DIV, O.IDIV: D_IDIV, O.MOD: D_MOD,
           O.POW: D_POW, O.SHL: D_SHL, O.SHR: D_SHR, O.BAND: D_BAND, O.BOR: D_BOR, O.BXOR: D_BXOR, O.BNOT: D_BNOT, O.LT: D_LT,
           O.GT: D_GT, O.LEQ: D_LEQ, O.GEQ: D_GEQ, O.EQ: D_EQ, O.NEQ: D_NEQ, O.LAND: D_LAND, O.LOR: D_LOR, O.LNOT: D_LNOT}


NATIVE_KINDS = {"mod_inv": 1, "ec_add": 2, "ec_double": 3, "long_div": 4}     # csrc/cw_call.hip.h eval_call_native / eval_call_long_div


def _encode_function(fn, cid, q):
    """rtcode bytecode -> device form (see the D_CALL comment at the top); constants go through the schedule's table"""
    from ..frontend import rtcode as R
    if fn["n_regs"] >= (1 << 16):
        raise ValueError("function %s needs too many registers" % fn["name"])

    def opnd(x, consts):
        if x is None:
            return 0
        return x[1] if x[0] == 'r' else (FN_CONST | cid(consts[x[1]] % q))

    out = np.zeros((len(fn["code"]), 4), dtype=np.uint32)
    consts = fn["consts"]
    for i, (op, d, a, b) in enumerate(fn["code"]):
        if op == R.F_RET:
            out[i] = (F_RET, 0, 0, 0)
        elif op == R.F_JMP:
            out[i] = (F_JMP, d, 0, 0)
        elif op == R.F_JZ:
            out[i] = (F_JZ, d, opnd(a, consts), 0)
        elif op == R.F_LDX:
            out[i] = (F_LDX, d, a, b[0] | (b[1] << 16))
        elif op == R.F_STX:
            out[i] = (F_STX, d, opnd(a, consts), b[0] | (b[1] << 16))
        else:
            out[i] = (_FN_ALU[op], d, opnd(a, consts), opnd(b, consts))
    # native closed form (circuits/bigint_func.py): the device computes the result directly when the modulus is a prime its
    # field code takes (225..256 bits, with n k bits = its limb capacity); otherwise the body is interpreted
    native = None
    nat = fn.get("native")
    if nat is not None:
        kind, n_, k_, modulus = nat
        if kind == "long_div":
            # (the tag's last field is m) the device routine takes limbs of 32..64 bits, a divisor of at most 256 and a
            # dividend of at most 640 bits
            m_ = modulus
            if 32 <= n_ <= 64 and 1 <= k_ <= 15 and 1 <= m_ <= 15 and n_ * k_ <= 256 and (k_ + m_) * n_ <= 640 \
                    and fn["n_args"] == 2 * k_ + m_ and fn["ret_base"] == fn["n_args"] and fn["n_ret"] == m_ + 1 + k_:
                native = (NATIVE_KINDS[kind], n_, k_, m_)
        elif 225 <= modulus.bit_length() <= 256 and modulus & 1 and n_ <= 64 and k_ <= 15 and n_ * k_ >= modulus.bit_length() and n_ * k_ <= 256 \
                and fn["ret_base"] == fn["n_args"]:
            native = (NATIVE_KINDS[kind], n_, k_, modulus)
    return fn["n_regs"], out, native


def _finish_pipe(fc, stream, dconsts, lconsts, lcid, witness_map, pipe, stats):
    import sys
    from . import pipe as PP
    n_signals = fc.n_signals
    n_before = len(lconsts)
    pl = PP.plan_pipe(stream, n_signals, dconsts, sys.modules[__name__], pipe[0], pipe[1])
    t = Tape()
    t.kind = 1
    t.pipe = (pipe[0], pipe[1], pl["ring"])
    t.prime = fc.prime
    t.q = fc.fp.q
    t.n_signals = n_signals
    t.n_tslots = pl["n_tslots"]
    t.rows = pl["rows"]
    t.stream_off = np.asarray([0, len(pl["rows"])], dtype=np.uint32)
    t.extras = pl["loads"].reshape(-1)
    t.extra_off = np.asarray([0, len(t.extras)], dtype=np.uint32)
    tt = np.zeros((len(pl["terms"]) + 4, 4), dtype=np.uint32)
    for j, (tk, te, cf, is_dotc) in enumerate(pl["terms"]):
        if is_dotc:
            tt[j] = (tk, te, lcid(cf), 0)
        else:
            m = abs(cf)
            tt[j] = (tk | (0x80000000 if cf < 0 else 0), te, m & 0xFFFFFFFF, m >> 32)
    assert len(lconsts) == n_before, "the limb-form constant table is shared by all variants"
    t.terms = tt
    t.term_off = np.asarray([0, len(pl["terms"])], dtype=np.uint32)
    t.lconsts = lconsts
    t.functions = []
    t.n_lds = pl["ring"] + 2 * pipe[1]
    t.n_strands = 1
    t.consts = dconsts
    if witness_map is None:
        witness_map = np.arange(n_signals, dtype=np.uint32)
    t.witness2signal = np.asarray(witness_map, dtype=np.uint32)
    t.n_witness = len(t.witness2signal)
    t.inputs = list(fc.inputs)
    t.main_input_start = fc.main_input_start
    t.n_main_inputs = fc.n_main_inputs
    t.n_pub_in = fc.n_pub_in
    dops = t.rows[:, 0] & 0xFF
    t.stats = dict(stats)
    t.stats.update(pl["stats"])
    t.stats.update({
        "rows": int(len(t.rows)), "strands": 1, "temp_slots": t.n_tslots, "consts": len(dconsts), "barriers": 0,
        "mmul": int((dops == D_MMUL).sum()), "mul2": int((dops == D_MUL2).sum()),
        "mulc": int(((dops == D_MULC) | (dops == D_MADDC)).sum()), "dotc": int((dops == D_DOTC).sum()),
        "addsub": int(((dops == D_ADD) | (dops == D_SUB) | (dops == D_NEG)).sum()), "inv": int((dops == D_INV).sum()),
        "linsum_terms": len(pl["terms"]),
    })
    return t


def lower(fc: FlatCircuit, witness_map=None, n_strands: int = 1, pipe=None, mont: bool = False, fuse_bits=None) -> Tape:
    """pipe = (rows per batch, loads per batch): lower to the pipelined single-wave variant (pipe.py) instead of strands.
    mont: signals in Montgomery form (pass A6; every variant of a circuit must use the same setting)
    fuse_bits: runs of bit rows as D_BITS rows (pass A7); None = where the interpreting kernel runs the schedule anyway
    (circuits with run-time functions)"""
    q = fc.fp.q
    if not 225 <= q.bit_length() <= 256:
        raise ValueError("hip_elements targets circom's 253..256-bit primes (4 x 64-bit limbs); prime %s has %d bits "
                         "(the 64-bit Goldilocks runtime is a separate code path of the reference, out of scope)"
                         % (fc.prime, q.bit_length()))
    circuit_signals = fc.n_signals
    if (fc.code["op"] == O.LOG).any():
        # log(...) arguments stay in the table as HIDDEN signals behind the circuit's own (flatten.code_with_log_copies):
        # the schedule computes and keeps them like signals, the witness list does not name them, cw_get_log reads them
        import copy
        from ..frontend.flatten import log_program, code_with_log_copies
        log_prog, log_strings = log_program(fc), list(fc.log_strings)
        fc = copy.copy(fc)
        fc.code, n_logv = code_with_log_copies(fc)
        fc.n_signals = circuit_signals + n_logv
        if witness_map is None:
            witness_map = np.arange(circuit_signals, dtype=np.uint32)
    else:
        log_prog, log_strings = [], []
    n_signals = fc.n_signals
    functions = getattr(fc, "functions", ())
    if pipe is not None:
        n_strands = 1
    if functions and (fc.code["op"] == O.CALL).any():
        if os.environ.get("CW_CALL_STRANDS", "1") == "0":
            n_strands = 1           # (experiments) tier-2 code in program order, one wave per 64 instances
        if pipe is not None:
            raise ValueError("circuits that call run-time functions have no pipelined variant")
    rows, dconsts, n_vtemps, cid, plain = _expand(fc)
    rows, n_vtemps, n_inv_batches = _batch_inversions(rows, n_vtemps, cid)
    rows, n_lin, n_bit = _fuse_linear(rows, plain, q, cid)
    n_bitsums = _fuse_linear.n_bitsum
    n_conv = 0
    if mont:
        if functions and (fc.code["op"] == O.CALL).any():
            raise ValueError("circuits that call run-time functions compute on canonical values")
        rows, n_vtemps, n_conv = _to_montgomery(rows, plain, cid, q, fc.fp.Rdev, n_vtemps)
    # limb-form constant table of the D_DOTC terms, interned in program order (identical for every strand count)
    lconsts, lconst_id = [], {}

    def lcid(c):
        v = (c % q) * fc.fp.Rdev % q
        i = lconst_id.get(v)
        if i is None:
            i = len(lconsts)
            lconst_id[v] = i
            lconsts.append(v)
        return i

    for r in rows:
        if r.op == D_DOTC:
            for tm in r.terms:
                lcid(tm[2])
    n_split = 0
    if n_strands > 1:      # one strand prefers the original chains (register forwarding, no extra temps)
        rows, n_vtemps = _reassociate(rows, n_vtemps)
        rows, n_vtemps, n_split = _split_linsums(rows, n_vtemps, n_strands)
    rows, n_elided = _alias(rows, n_signals)
    n_madd = 0
    if n_strands == 1:
        rows, n_madd = _fuse_madd(rows)
    n_bits_fused = 0
    has_calls = bool(functions) and any(r.op == D_CALL for r in rows)
    if fuse_bits is None:
        # only schedules that the interpreting kernel runs anyway (the emitted code has no body for the row)
        # (a single-strand schedule of such a circuit still has an emitted form, fpjit.py, which has no body for the row)
        fuse_bits = has_calls and n_strands > 1 and pipe is None and os.environ.get("CW_FUSE_BITS", "1") != "0"
    if fuse_bits:
        if n_signals >= X_NEXT:
            raise ValueError("bit-field rows need signal numbers below 2^29")
        rows, n_bits_fused = _fuse_bits(rows)
    streams, n_levels, forced_full = _schedule(rows, n_signals, n_strands, functions, interp_costs=has_calls)
    if n_bits_fused:
        # a row that takes a bit of the D_BITS row right in front of it as a / b / c operand would have requested it before
        # the bits were stored (operands are fetched one row ahead): a spacer row in between
        for si, st in enumerate(streams):
            out_, last = [], None
            for it in st:
                if it != "B" and last is not None and any((k, v) in last for k, v in ((it.ak, it.av), (it.bk, it.bv), (it.ck, it.cv))):
                    sp = _Row(D_ALSO, KD_NONE, 0, K_SIG, 0)
                    sp.strand = si
                    out_.append(sp)
                out_.append(it)
                last = {(K_SIG, dv) for dv, _ in it.multi} if (it != "B" and it.multi) else None
            streams[si] = out_
    multi = n_strands > 1
    if pipe is not None:
        t = _finish_pipe(fc, streams[0], dconsts, lconsts, lcid, witness_map, pipe,
                         {"copies_elided": n_elided, "fused_madd": n_madd, "inv_batches": n_inv_batches, "linsum": n_lin,
                          "bit": n_bit, "asserts_proved": getattr(_expand, "n_proved", 0), "mont": int(mont), "mont_conversions": n_conv})
        t.mont = bool(mont)
        t.log_strings, t.log_prog = log_strings, log_prog
        return t

    def vid(k, v):
        return v if k == K_SIG else n_signals + v

    # ---- pass D1: walk the streams; find PREV-forwarded operands, cross-strand flows, liveness ---------------------
    # time unit: row position for one strand, barrier epoch for several
    prod_strand, def_time = {}, {}
    bits_vals = set()   # values stored by D_BITS rows (no register, no LDS copy)
    plan = []           # per stream: list of [row | 'B', (prev_a, prev_b), time]
    for si, s in enumerate(streams):
        prev_val = None
        epoch = 0
        items = []
        for pos, r in enumerate(s):
            if r == "B":
                epoch += 1
                items.append(["B", None, epoch])
                continue
            t = epoch if multi else pos
            fl = tuple(prev_val is not None and (k, v) == prev_val for k, v in ((r.ak, r.av), (r.bk, r.bv)))
            if r.terms:     # per-term PREV flags ride along as a third element
                fl = fl + (tuple(prev_val is not None and (tm[0], tm[1]) == prev_val for tm in r.terms),)
            items.append([r, fl, t])
            if r.op not in _NO_VALUE:
                prev_val = (r.dk, r.dv) if r.dk in (K_SIG, K_TMP) else None
                if r.dk in (K_SIG, K_TMP):
                    prod_strand[vid(r.dk, r.dv)] = si
                    def_time[vid(r.dk, r.dv)] = t
                for dv_, _ in (r.multi or ()):        # D_BITS: every bit is a value of this strand, in the value table only
                    prod_strand[vid(K_SIG, dv_)] = si
                    def_time[vid(K_SIG, dv_)] = t
                    bits_vals.add(vid(K_SIG, dv_))
        plan.append(items)
    # Cross-strand flows.  A value produced in epoch d by one strand and read by another must be visible to the
    # reader: either through an LDS slot (LIGHT barriers suffice) or through the value table, which needs a FULL
    # barrier (vmcnt(0)) between the store and the load.  Every FULL_PERIOD-th barrier is FULL anyway, so reads that
    # happen after the next periodic FULL barrier go through the value table for free; only the short-range reads
    # before it need an LDS slot (and only until that barrier: bounded LDS lifetime).
    mem_last = {}       # value id -> last time it is read from the value table
    x_uses = {}         # value id -> sorted epochs at which ANOTHER strand reads it
    call_regs = set()   # registers of function calls: pinned slots of the value table, read and written through memory only
    for r in rows:
        if r.op == D_CALL:
            call_regs.update(vid(K_TMP, r.bv + k) for k in range(functions[r.av]["n_regs"]))
    for si, items in enumerate(plan):
        for r, fl, t in items:
            if r == "B":
                continue
            if r.op == D_CALL:
                # the arguments are read from the value table whatever strand stored them (FULL barriers around the call's
                # level, _schedule): never an LDS hand-off, never forwarded
                for tm in r.terms:
                    x = vid(tm[0], tm[1])
                    mem_last[x] = max(mem_last.get(x, -1), t)
                continue
            ops = [(r.ak, r.av, fl[0]), (r.bk, r.bv, fl[1]), (r.ck, r.cv, False)]
            if r.terms:
                ops.extend((tm[0], tm[1], pf) for tm, pf in zip(r.terms, fl[2]))
            for k, v, is_prev in ops:
                if k not in (K_SIG, K_TMP) or is_prev:
                    continue
                x = vid(k, v)
                if multi and x in prod_strand and prod_strand[x] != si and x not in call_regs:
                    x_uses.setdefault(x, []).append(t)
                else:
                    mem_last[x] = max(mem_last.get(x, -1), t)

    # ---- pass D2: LDS slots for short-range cross-strand reads; which barriers are FULL ------------------------------
    n_lds = lds_slots_for(n_strands)
    lds_of, lds_until = {}, {}
    full_after = set()          # epochs whose closing barrier must wait for global stores (vmcnt(0))
    n_lds_used = 0
    if multi:
        K = FULL_PERIOD
        n_epochs = n_levels + 1
        full_after.update(e for e in range(n_epochs) if (e + 1) % K == 0)
        full_after.update(forced_full)          # the barriers around a function call's level

        def next_full(d):       # first epoch >= d whose closing barrier is periodic-FULL
            return d + (K - 1 - d % K)

        free = list(range(n_lds - 1, -1, -1))
        release = {}            # epoch -> slots that become reusable once that epoch is over
        cur = -1
        for t, x in sorted((def_time[x], x) for x in x_uses):
            while cur < t - 1:
                cur += 1
                free.extend(release.pop(cur, ()))
            b = next_full(t)
            near = [u for u in x_uses[x] if u <= b]
            far = [u for u in x_uses[x] if u > b]
            if far:
                mem_last[x] = max(mem_last.get(x, -1), max(far))
            if not near:
                continue
            if x in bits_vals:                          # a bit of a D_BITS row: through the value table, behind a FULL barrier
                full_after.add(t)
                mem_last[x] = max(mem_last.get(x, -1), max(near))
                continue
            if free:
                sl = free.pop()
                lds_of[x] = sl
                lds_until[x] = max(near)
                n_lds_used = max(n_lds_used, sl + 1)
                release.setdefault(max(near), []).append(sl)
            else:
                full_after.add(t)                       # hand-off through the value table right away
                mem_last[x] = max(mem_last.get(x, -1), max(near))

    # ---- pass D3: global temp slots (only temps that are actually read from global memory) ------------------------
    slot_of = {}
    n_tslots = 0
    pinned = set()
    for r in rows:                       # register windows of function calls: consecutive slots, never reused
        if r.op == D_CALL:
            for k in range(functions[r.av]["n_regs"]):
                slot_of[r.bv + k] = n_tslots + k
                pinned.add(r.bv + k)
            n_tslots += functions[r.av]["n_regs"]
    tmp_last = {x - n_signals: t for x, t in mem_last.items() if x >= n_signals}
    if multi:
        by_time = sorted((def_time[n_signals + v], v) for v in tmp_last if v not in pinned)
        free_at, free, cur = {}, [], -1
        for t, v in by_time:
            while cur < t - 1:
                cur += 1
                free.extend(free_at.pop(cur, ()))
            if free:
                sl = free.pop()
            else:
                sl = n_tslots
                n_tslots += 1
            slot_of[v] = sl
            free_at.setdefault(tmp_last[v], []).append(sl)
    else:
        free = []
        release = {}
        owner = {}                              # slot -> the temporary it currently holds
        for v, t in tmp_last.items():
            release.setdefault(t, []).append(v)
        for (r, fl, t) in plan[0]:
            for v in release.get(t, ()):        # operands are read before the destination is written
                if v in slot_of and v not in pinned and owner.get(slot_of[v]) == v:
                    free.append(slot_of[v])
                    del owner[slot_of[v]]
            if r != "B" and r.dk == K_TMP and r.dv in tmp_last and r.dv not in pinned:
                # a temporary written more than once (`var` of a `<--` computation: the trace re-targets it) keeps its slot
                # while it is live: a second slot for the later value would leave the first one owned by nobody and free
                # it under a value that is still going to be read
                if r.dv in slot_of and owner.get(slot_of[r.dv]) == r.dv:
                    continue
                if free:
                    sl = free.pop()
                else:
                    sl = n_tslots
                    n_tslots += 1
                slot_of[r.dv] = sl
                owner[sl] = r.dv

        # the allocation is checked by replaying the slots' contents: every table read of a temporary must find that very
        # temporary in its slot (a clobbered slot is a wrong witness that only SOME inputs reveal)
        content = {}
        for (r, fl, t) in plan[0]:
            if r == "B":
                continue
            ops = [(r.ak, r.av, fl[0]), (r.bk, r.bv, fl[1]), (r.ck, r.cv, False)]
            if r.terms:
                ops.extend((tm[0], tm[1], pf) for tm, pf in zip(r.terms, fl[2]))
            for k, v, is_prev in ops:
                if k == K_TMP and not is_prev and v in slot_of and v not in pinned:
                    if content.get(slot_of[v]) != v:
                        if __import__("os").environ.get("CW_SLOT_DEBUG"):
                            print("DEBUG tmp", v, "def_time", def_time.get(n_signals + v), "tmp_last", tmp_last.get(v), "read at", t,
                                  "clobberer", content.get(slot_of[v]), "its def", def_time.get(n_signals + content.get(slot_of[v], 0)))
                        raise AssertionError("temp slot allocation: slot %d holds temporary %s when temporary %d is read at row %d (op %d)"
                                             % (slot_of[v], content.get(slot_of[v]), v, t, r.op))
            if r.dk == K_TMP and r.dv in slot_of and r.dv not in pinned:
                content[slot_of[r.dv]] = r.dv
            for exl in ([r.extra] if r.extra else []) + [x for _, x in (r.multi or ()) if x]:
                for xk, xv in exl:
                    if xk == K_TMP and xv in slot_of and xv not in pinned:
                        content[slot_of[xv]] = xv

    # ---- pass E: encode ------------------------------------------------------------------------------------------------

    enc = []
    extras = []
    terms = []          # (kind, index, signed coefficient | limb-constant index) in stream/row order
    stream_off = [0]
    extra_off = [0]
    term_off = [0]
    seqs, seq_off = [], [0]   # flat-operation index of every row that can fail (in stream order), per strand
    n_prev = n_ldsops = 0
    for si, items in enumerate(plan):
        for (r, fl, t) in items:
            if r == "B":
                enc.append((D_BARRIER, 1 if (t - 1) in full_after else 0, 0, 0))
                continue

            def o_enc(k, v, is_prev):
                if is_prev:
                    return KO_PREV, 0
                if k in (K_SIG, K_TMP):
                    x = vid(k, v)
                    if x in lds_of and prod_strand.get(x) != si and t <= lds_until[x]:
                        return K_LDS, lds_of[x]
                    return (K_TMP, slot_of[v]) if k == K_TMP else (K_SIG, v)
                if k == K_NONE:
                    return 0, 0
                return k, v

            if r.op == D_LINSUM or r.op == D_DOTC:
                ka, va = 0, len(r.terms)
                kb, vb = (K_CONST, r.bv) if r.bk == K_CONST else (0, 0)
                for tm, pf in zip(r.terms, fl[2]):
                    tk, tv = o_enc(tm[0], tm[1], pf)
                    if r.op == D_DOTC:
                        terms.append((tk, tv, lcid(tm[2])))       # index of coef*R' in the limb-form constant table
                    else:
                        terms.append((tk, tv, tm[2]))
                    n_prev += pf
            elif r.op == D_BIT or r.op == D_BITS:
                ka, va = o_enc(r.ak, r.av, fl[0])
                kb, vb = 0, r.bv
                n_prev += fl[0]
            elif r.op == D_CALL:
                ka, va = 0, r.av                          # function id
                kb, vb = K_TMP, slot_of[r.bv]             # first register slot
            else:
                ka, va = o_enc(r.ak, r.av, fl[0])
                kb, vb = o_enc(r.bk, r.bv, fl[1])
                n_prev += fl[0] + fl[1]
            n_ldsops += (ka == K_LDS) + (kb == K_LDS)
            ex = []
            if r.dk == K_TMP:
                kd, vd = (K_TMP, slot_of[r.dv]) if r.dv in slot_of else (KD_NONE, 0)
            elif r.dk == K_SIG:
                kd, vd = K_SIG, r.dv
            else:
                kd, vd = KD_NONE, 0
            if r.dk in (K_SIG, K_TMP) and vid(r.dk, r.dv) in lds_of:
                if kd == KD_NONE:
                    kd, vd = K_LDS, lds_of[vid(r.dk, r.dv)]
                else:
                    ex.append(X_LDS | lds_of[vid(r.dk, r.dv)])
            if r.extra:
                for k, v in r.extra:
                    if k == K_SIG:
                        ex.append(v)
                    elif v in slot_of:
                        ex.append(X_TMP | slot_of[v])
            if r.multi:                                  # D_BITS: per bit its signal, then its copies; X_NEXT starts a new bit
                for j, (dv_, exl) in enumerate(r.multi):
                    ex.append(dv_ | (X_NEXT if j else 0))
                    for k, v in (exl or ()):
                        if k == K_SIG:
                            ex.append(v)
                        elif v in slot_of:
                            ex.append(X_TMP | slot_of[v])
            assert len(ex) <= MAX_EXTRA, "fan-out of one value exceeds the extra-destination field"
            if r.op in _FAIL_OPS:
                seqs.append(r.seq)
            enc.append((r.op | (kd << SH_DK) | (ka << SH_AK) | (kb << SH_BK) | (len(ex) << SH_NX) | (r.flag << SH_FLAG),
                        vd, va, vb))
            extras.extend(ex)
        stream_off.append(len(enc))
        extra_off.append(len(extras))
        term_off.append(len(terms))
        seq_off.append(len(seqs))
    out = np.asarray(enc, dtype=np.uint32).reshape(-1, 4)

    t = Tape()
    t.prime = fc.prime
    t.q = q
    t.n_signals = n_signals
    t.n_tslots = n_tslots
    t.rows = out
    t.stream_off = np.asarray(stream_off, dtype=np.uint32)
    t.extras = np.asarray(extras + [0, 0, 0, 0], dtype=np.uint32)    # padded: the kernel always reads 4 ahead
    t.extra_off = np.asarray(extra_off, dtype=np.uint32)
    # term table: 4 x u32 per term = kind, index, |coef| lo, |coef| hi with the sign in bit 31 of the kind word
    tt = np.zeros((len(terms) + 4, 4), dtype=np.uint32)      # padded: the kernel reads terms four at a time
    for j, (tk, tv, cf) in enumerate(terms):
        m = abs(cf)
        tt[j] = (tk | (0x80000000 if cf < 0 else 0), tv, m & 0xFFFFFFFF, m >> 32)
    t.terms = tt
    t.term_off = np.asarray(term_off, dtype=np.uint32)
    t.seqs = np.asarray(seqs, dtype=np.uint32)
    t.seq_off = np.asarray(seq_off, dtype=np.uint32)
    t.lconsts = lconsts
    t.functions = [_encode_function(f, cid, q) for f in functions]
    t.n_lds = n_lds_used
    t.n_strands = n_strands
    t.mont = bool(mont)
    t.consts = dconsts
    t.n_dat_consts = len(fc.constants)           # what the .dat holds (the reference's constant list), not the schedule's table
    t.n_io_templates = len(getattr(fc, "io_map", ()))
    if witness_map is None:
        witness_map = np.arange(n_signals, dtype=np.uint32)     # --O0: identity
    t.witness2signal = np.asarray(witness_map, dtype=np.uint32)
    t.n_witness = len(t.witness2signal)
    t.inputs = list(fc.inputs)
    t.main_input_start = fc.main_input_start
    t.n_main_inputs = fc.n_main_inputs
    t.n_pub_in = fc.n_pub_in
    dops = out[:, 0] & 0xFF
    t.stats = {
        "rows": int(len(out)),
        "mmul": int((dops == D_MMUL).sum()),
        "addsub": int(((dops == D_ADD) | (dops == D_SUB) | (dops == D_NEG)).sum()),
        "copy": int((dops == D_COPY).sum()),
        "copies_elided": n_elided,
        "asserts_proved": getattr(_expand, "n_proved", 0),
        "extra_dsts": len(extras),
        "prev_operands": n_prev,
        "lds_operands": n_ldsops,
        "lds_slots": n_lds_used,
        "mul2": int((dops == D_MUL2).sum()),
        "mulc": int(((dops == D_MULC) | (dops == D_MADDC)).sum()),
        "linsum": n_lin, "linsum_terms": len(terms), "bit": n_bit, "dotc": int((dops == D_DOTC).sum()),
        "madd": int((dops == D_MADD).sum()),
        "fused_madd": n_madd,
        "inv": int((dops == D_INV).sum()),
        "inv_batches": n_inv_batches,
        "bits_rows": int((dops == D_BITS).sum()), "bits_fused": n_bits_fused, "bit_sums_folded": n_bitsums,
        "pow2_divisions": getattr(_expand, "n_pow2", 0),
        "linsum_splits": n_split,
        "barriers": n_levels,
        "full_barriers": len(full_after),
        "mont": int(mont), "mont_conversions": n_conv,
        "strands": n_strands,
        "temp_slots": n_tslots,
        "consts": len(dconsts),
    }
    t.log_strings, t.log_prog = log_strings, log_prog
    return t
