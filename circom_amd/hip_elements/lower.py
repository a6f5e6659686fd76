"""hip_elements lowering: flattened circuit -> batched signal-evaluation schedule ("tape").

This is the new back-end the north-star adds beside `c_elements`/`wasm_elements`
(code_producers/src/c_elements/mod.rs:6-39 is the producer it parallels): instead of emitting a
per-template C++/WASM program that interprets one input, it emits ONE straight-line schedule over
global value slots that the fixed HIP kernels (circom_amd/csrc/) evaluate for thousands of
instances at once (outer loop = schedule step, inner = instance/lane).

Representation policy ("N policy"): every value slot holds the CANONICAL residue in [0,q) as
8 x u32 limbs.  Rationale (vs. the reference's tagged short/long/Montgomery union, fr.hpp:17-21):
  * `.wtns` stores canonical values (main.cpp:326-332) -> no egress conversion pass,
  * bitwise/relational/shift operators are defined on canonical values,
  * add/sub are representation-agnostic,
  * a product with a compile-time constant is ONE raw Montgomery multiplication when the constant
    is pre-scaled by R here (MMUL(x, c*R) = x*c), and a product of two run-time values is
    MMUL(MMUL(x,y), R^2).
Raw device ops therefore include MMUL (a*b*R^-1 mod q) and the lowering owns all scaling.

Schedule rows are 4 x u32: w0 = op | dk<<8 | ak<<10 | bk<<12, then dst, a, b.  Operand kinds:
0 = signal slot, 1 = temp slot, 2 = constant-table index.  SELECT carries its third operand in a
following EXT row.
"""
from __future__ import annotations

import numpy as np

from .. import opcodes as O
from ..frontend.flatten import FlatCircuit

# device opcodes (csrc/cw_tape.h must match)
(D_COPY, D_ADD, D_SUB, D_NEG, D_MMUL, D_INV, D_IDIV, D_MOD, D_POW, D_SHL, D_SHR, D_BAND, D_BOR, D_BXOR,
 D_BNOT, D_LT, D_GT, D_LEQ, D_GEQ, D_EQ, D_NEQ, D_LAND, D_LOR, D_LNOT, D_SELECT, D_EXT, D_ASSERT_EQ,
 D_ASSERT_NZ) = range(28)
D_NAMES = ["copy", "add", "sub", "neg", "mmul", "inv", "idiv", "mod", "pow", "shl", "shr", "band", "bor",
           "bxor", "bnot", "lt", "gt", "leq", "geq", "eq", "neq", "land", "lor", "lnot", "select", "ext",
           "assert_eq", "assert_nz"]

K_SIG, K_TMP, K_CONST, K_NONE = O.K_SIG, O.K_TMP, O.K_CONST, O.K_NONE

_DIRECT = {O.COPY: D_COPY, O.ADD: D_ADD, O.SUB: D_SUB, O.NEG: D_NEG, O.IDIV: D_IDIV, O.MOD: D_MOD,
           O.POW: D_POW, O.SHL: D_SHL, O.SHR: D_SHR, O.BAND: D_BAND, O.BOR: D_BOR, O.BXOR: D_BXOR,
           O.BNOT: D_BNOT, O.LT: D_LT, O.GT: D_GT, O.LEQ: D_LEQ, O.GEQ: D_GEQ, O.EQ: D_EQ, O.NEQ: D_NEQ,
           O.LAND: D_LAND, O.LOR: D_LOR, O.LNOT: D_LNOT, O.ASSERT_EQ: D_ASSERT_EQ, O.ASSERT_NZ: D_ASSERT_NZ}


class Tape:
    """The lowered schedule + tables, ready to serialise (writers.py) or hand to the runtime."""

    def __init__(self):
        self.prime = ""
        self.q = 0
        self.n_signals = 0
        self.n_tslots = 0
        self.n_witness = 0
        self.rows = None            # (n,4) uint32
        self.consts = []            # raw residues (python ints)
        self.witness2signal = None  # uint32[n_witness]
        self.inputs = []            # (name, slot, size)
        self.main_input_start = 0
        self.n_main_inputs = 0
        self.stats = {}
        self.rbits = 261            # Montgomery radix exponent of MMUL rows


def _dce(code, n_temps):
    op = code["op"]
    n = len(op)
    keep = np.zeros(n, dtype=bool)
    live = np.zeros(max(n_temps, 1), dtype=bool)
    dk, dv = code["dk"], code["dv"]
    cols = ((code["ak"], code["av"]), (code["bk"], code["bv"]), (code["ck"], code["cv"]))
    for i in range(n - 1, -1, -1):
        o = op[i]
        need = (o == O.ASSERT_EQ or o == O.ASSERT_NZ or dk[i] == K_SIG or (dk[i] == K_TMP and live[dv[i]]))
        if need:
            keep[i] = True
            for kk, vv in cols:
                if kk[i] == K_TMP:
                    live[vv[i]] = True
    return keep


def lower(fc: FlatCircuit, witness_map=None) -> Tape:
    fp = fc.fp
    q = fp.q
    code = fc.code
    keep = _dce(code, fc.n_temps)
    idx = np.nonzero(keep)[0]
    op = code["op"][idx].tolist()
    dk = code["dk"][idx].tolist(); dv = code["dv"][idx].tolist()
    ak = code["ak"][idx].tolist(); av = code["av"][idx].tolist()
    bk = code["bk"][idx].tolist(); bv = code["bv"][idx].tolist()
    ck = code["ck"][idx].tolist(); cv = code["cv"][idx].tolist()
    consts_in = fc.constants

    dconsts = []
    dconst_id = {}

    def cid(v):
        i = dconst_id.get(v)
        if i is None:
            i = len(dconsts)
            dconst_id[v] = i
            dconsts.append(v)
        return i

    R, R2 = fp.Rdev, fp.Rdev2        # device radix R' = 2^261
    # virtual temps: flat temp ids, plus fresh ones for expansion intermediates
    next_tmp = [fc.n_temps]

    def fresh():
        t = next_tmp[0]
        next_tmp[0] += 1
        return t

    rows = []   # (dop, dk, dv, ak, av, bk, bv) with virtual temps; consts already device ids

    def opnd(k, v, scale=1):
        """flat operand -> device operand; constants become device-constant ids (optionally pre-scaled)."""
        if k == K_CONST:
            return K_CONST, cid((consts_in[v] * scale) % q)
        return k, v

    n_mmul = 0
    for i in range(len(op)):
        o = op[i]
        if o == O.MUL:
            a_c, b_c = ak[i] == K_CONST, bk[i] == K_CONST
            if a_c or b_c:
                # x * c  ->  MMUL(x, c*R)
                if a_c:
                    xk, xv = bk[i], bv[i]
                    c = consts_in[av[i]]
                else:
                    xk, xv = ak[i], av[i]
                    c = consts_in[bv[i]]
                rows.append((D_MMUL, dk[i], dv[i], xk, xv, K_CONST, cid((c * R) % q)))
                n_mmul += 1
            else:
                t = fresh()
                rows.append((D_MMUL, K_TMP, t, ak[i], av[i], bk[i], bv[i]))
                rows.append((D_MMUL, dk[i], dv[i], K_TMP, t, K_CONST, cid(R2)))
                n_mmul += 2
        elif o == O.DIV:
            # a / b = a * inv(b); inv(0) = 0 (generic/fr.cpp:2895-2912)
            t = fresh()
            kb, vb = opnd(bk[i], bv[i])
            rows.append((D_INV, K_TMP, t, kb, vb, K_NONE, 0))
            if ak[i] == K_CONST:
                rows.append((D_MMUL, dk[i], dv[i], K_TMP, t, K_CONST, cid((consts_in[av[i]] * R) % q)))
                n_mmul += 1
            else:
                t2 = fresh()
                rows.append((D_MMUL, K_TMP, t2, ak[i], av[i], K_TMP, t))
                rows.append((D_MMUL, dk[i], dv[i], K_TMP, t2, K_CONST, cid(R2)))
                n_mmul += 2
        elif o == O.SELECT:
            ka, va = opnd(ak[i], av[i])
            kb, vb = opnd(bk[i], bv[i])
            kc, vc = opnd(ck[i], cv[i])
            rows.append((D_SELECT, dk[i], dv[i], ka, va, kb, vb))
            rows.append((D_EXT, K_NONE, 0, kc, vc, K_NONE, 0))
        else:
            d = _DIRECT[o]
            ka, va = opnd(ak[i], av[i])
            kb, vb = opnd(bk[i], bv[i]) if bk[i] != K_NONE else (K_NONE, 0)
            rows.append((d, dk[i], dv[i], ka, va, kb, vb))

    # ---- temp slot allocation (linear scan over virtual temps) ---------------------------------
    nrows = len(rows)
    last_use = {}
    for r in range(nrows):
        _, _, _, ka, va, kb, vb = rows[r]
        if ka == K_TMP:
            last_use[va] = r
        if kb == K_TMP:
            last_use[vb] = r
    free = []
    slot_of = {}
    n_tslots = 0
    out = np.zeros((nrows, 4), dtype=np.uint32)
    for r in range(nrows):
        d, kd, vd, ka, va, kb, vb = rows[r]
        sa = slot_of[va] if ka == K_TMP else va
        sb = slot_of[vb] if kb == K_TMP else vb
        # release operands whose last use is this row *before* allocating dst: dst may reuse the slot
        # (kernels read all operands before writing)
        if ka == K_TMP and last_use.get(va) == r:
            free.append(slot_of.pop(va))
        if kb == K_TMP and vb != va and last_use.get(vb) == r and vb in slot_of:
            free.append(slot_of.pop(vb))
        if kd == K_TMP:
            if vd in last_use:
                if free:
                    s = free.pop()
                else:
                    s = n_tslots
                    n_tslots += 1
                slot_of[vd] = s
                sd = s
            else:
                sd = 0  # dead temp cannot happen after DCE except for SELECT/EXT pairs
        else:
            sd = vd
        kd_ = 0 if kd == K_NONE else kd
        ka_ = 0 if ka == K_NONE else ka
        kb_ = 0 if kb == K_NONE else kb
        out[r, 0] = d | (kd_ << 8) | (ka_ << 10) | (kb_ << 12)
        out[r, 1] = sd
        out[r, 2] = sa
        out[r, 3] = sb

    t = Tape()
    t.prime = fc.prime
    t.q = q
    t.n_signals = fc.n_signals
    t.n_tslots = n_tslots
    t.rows = out
    t.consts = dconsts
    if witness_map is None:
        witness_map = np.arange(fc.n_signals, dtype=np.uint32)     # --O0: identity
    t.witness2signal = np.asarray(witness_map, dtype=np.uint32)
    t.n_witness = len(t.witness2signal)
    t.inputs = list(fc.inputs)
    t.main_input_start = fc.main_input_start
    t.n_main_inputs = fc.n_main_inputs
    dops = out[:, 0] & 0xFF
    t.stats = {
        "rows": nrows,
        "mmul": int((dops == D_MMUL).sum()),
        "addsub": int(((dops == D_ADD) | (dops == D_SUB) | (dops == D_NEG)).sum()),
        "copy": int((dops == D_COPY).sum()),
        "inv": int((dops == D_INV).sum()),
        "temp_slots": n_tslots,
        "consts": len(dconsts),
    }
    return t
