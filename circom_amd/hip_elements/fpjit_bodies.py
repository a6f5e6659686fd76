"""hip_elements, emitted 256-bit code, part 1: the ROW BODIES.

The reference's C back-end prints one C++ function per template and lets the C++ compiler turn every `Fr_mul(&a, &b, &c)`
into a call (compiler/src/circuit_design/template.rs:174-474, compute_bucket.rs:315-421).  The emitted 256-bit code of
`fpjit.py` has the same shape on the device: the row stream of the schedule becomes straight-line gfx950 code that loads
the operands, CALLS the operator and stores the result - and the operators it calls are the row bodies of this module.

A body is compiled from the very C++ the interpreting kernels use (`csrc/fp256.hip.h`, `csrc/cw_rowops.hip.h`), but with a
register interface fixed by this module instead of by a calling convention: the body is written as a kernel whose inputs
are DEFINED by an inline-asm statement with physical-register constraints (`"={v0}"(a.v[0])`) and whose outputs are
CONSUMED by another one (`"{v16}"(d.v[0])`); hipcc compiles the arithmetic in between with whatever registers are left;
the text between the two markers is cut out of the compiler's assembly and becomes a leaf routine that ends in
`s_setpc_b64`.  Registers the caller owns while the body runs (the operands being prefetched for the next row, the status
word, the lane offsets, the field parameters) are kept live THROUGH the body by the same two statements, so the compiler
cannot touch them.  Nothing here is measured or validated on trust: `parse_bodies` checks that the text between the markers
contains no memory instruction, no scratch, no call, and writes no register outside its budget.

Register map (wave64, at most 128 VGPRs so that 16 strands fit a workgroup):
  v[0:7] / v[8:15]     operands A, B of EVEN rows          v[24:31] / v[32:39]  operands A, B of ODD rows
  v[16:23]             D: result of the row = PREV of the next           v[40:75]   temporaries
  v[76:83]             G: running result of a LINSUM / DOTC row          v[84:119]  ACC: 18 x 64-bit columns (DOTC), 2 x 192 bits (LINSUM)
  v[120:127]           owned by the emitted code (fpjit.py)
  s[0:35] temporaries | s[36:37] 64-bit scalar argument | s[38:39] SELECT mask | s[24:32] DOTC coefficient limbs
  s[40:92] FpParams | s[93:101] owned by the emitted code (s[98:99] = return address)
"""
from __future__ import annotations

import hashlib
import os
import pickle
import re
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
CACHE_DIR = os.path.join(os.path.dirname(_HERE), "lib")

A_E, B_E, D_REG, A_O, B_O = 0, 8, 16, 24, 32
G_REG, ACC_REG = 76, 84
V_OWNED = list(range(120, 128))
KA_CONSTS, KA_FCODE, KA_FTAB = 31, 32, 33       # 64-bit words of the emitted kernel's argument block (fpjit.KERNARG_BYTES)
V_FB = 123                    # smallest index of a constraint the fused R1CS check found violated (0xFFFFFFFF: none)
S_ARG, S_SEL, S_COEF, S_PARAMS, S_OWNED = 36, 38, 24, 40, 93
S_RET = 98
N_VGPR = 128

# FpParams fields in declaration order (cw_tape.h) -> s[40 + k]
_P_FIELDS = [("q", 8), ("half", 8), ("r2", 8), ("one_m", 8), ("q29", 9), ("r2_29", 9), ("np29", 0), ("qbits", 0), ("topmask", 0)]
N_PARAM_SGPRS = sum(max(n, 1) for _, n in _P_FIELDS)       # 53


def _p_pins():
    out, s = [], S_PARAMS
    for name, n in _P_FIELDS:
        if n == 0:
            out.append((s, "P.%s" % name))
            s += 1
        else:
            for j in range(n):
                out.append((s, "P.%s[%d]" % (name, j)))
                s += 1
    return out


class Body:
    def __init__(self, name, code, vin=(), vout=(), sin=(), sout=(), keep_d=False, acc=False, g=False, parity=None, chk=False):
        self.chk = chk
        self.name, self.code = name, code
        self.vin, self.vout, self.sin, self.sout = list(vin), list(vout), list(sin), list(sout)
        self.keep_d, self.acc, self.g, self.parity = keep_d, acc, g, parity
        self.text = None          # instruction lines
        self.n_instr = 0
        self.vwritten = set()
        self.scratch, self.scratch_bytes = False, 0
        self.n_agpr = 0


def _fe(reg, var):
    return [(reg + k, "%s.v[%d]" % (var, k)) for k in range(8)]


def _specs():
    """Every body, in both parities where it has table operands.  `code` is C++ over: fe a, b (operands), fe d (result,
    read = PREV), uint32_t st, uint64_t sarg, uint64_t selmask, FpParams P; ACC bodies also see uint64_t acc[18] / Acc192
    pos, neg and fe g."""
    S = []
    two = {
        "add": "d = fe_add(a, b, P);", "sub": "d = fe_sub(a, b, P);", "mmul": "d = fe_mmul(a, b, P);",
        "mul2": "d = fe_mul2_auto(a, b, P);", "shl": "d = fe_shl(a, b, P);", "shr": "d = fe_shr(a, b, P);",
        "band": "d = fe_band(a, b, P);", "bor": "d = fe_bor(a, b, P);", "bxor": "d = fe_bxor(a, b, P);",
        "lt": "d = fe_small(fe_lt(a, b, P));", "gt": "d = fe_small(fe_lt(b, a, P));",
        "leq": "d = fe_small(!fe_lt(b, a, P));", "geq": "d = fe_small(!fe_lt(a, b, P));",
        "eq": "d = fe_small(fe_eq(a, b));", "neq": "d = fe_small(!fe_eq(a, b));",
        "land": "d = fe_small(!fe_is_zero(a) & !fe_is_zero(b));", "lor": "d = fe_small(!fe_is_zero(a) | !fe_is_zero(b));",
    }
    one = {"neg": "d = fe_neg(a, P);", "bnot": "d = fe_bnot(a, P);", "lnot": "d = fe_small(fe_is_zero(a));"}
    for par, (ra, rb) in (("e", (A_E, B_E)), ("o", (A_O, B_O))):
        for n, c in two.items():
            S.append(Body("%s_%s" % (n, par), c, vin=_fe(ra, "a") + _fe(rb, "b"), vout=_fe(D_REG, "d"), parity=par))
        for n, c in one.items():
            S.append(Body("%s_%s" % (n, par), c, vin=_fe(ra, "a"), vout=_fe(D_REG, "d"), parity=par))
        # d = mmul(a, b) + PREV
        S.append(Body("madd_%s" % par, "d = fe_add(fe_mmul(a, b, P), d, P);", vin=_fe(ra, "a") + _fe(rb, "b") + _fe(D_REG, "d"),
                      vout=_fe(D_REG, "d"), parity=par))
        # products by a compile-time constant: b = c R' (generic path), sarg = |val(c)| when the constant is small
        for cs, nm in ((0, "mulc0"), (1, "mulcp"), (2, "mulcn")):
            call = "fe_mulc_auto(a, b, %s, sarg, %s, P)" % ("true" if cs else "false", "true" if cs == 2 else "false")
            S.append(Body("%s_%s" % (nm, par), "d = %s;" % call, vin=_fe(ra, "a") + _fe(rb, "b"), vout=_fe(D_REG, "d"),
                          sin=[(S_ARG, "sarg")] if cs else [], parity=par))
            S.append(Body("%sa_%s" % (nm, par), "d = fe_add(%s, d, P);" % call, vin=_fe(ra, "a") + _fe(rb, "b") + _fe(D_REG, "d"),
                          vout=_fe(D_REG, "d"), sin=[(S_ARG, "sarg")] if cs else [], parity=par))
        # predication: SELECT latches the lanes whose condition holds, EXT picks per lane
        S.append(Body("select_%s" % par, "selmask = __ballot(!fe_is_zero(a));", vin=_fe(ra, "a"), sout=[(S_SEL, "selmask")],
                      keep_d=True, parity=par))
        S.append(Body("ext_%s" % par, "{ uint32_t ln;      // the lane number, by a volatile statement: NOT threadIdx (a body sees none of the kernel's entry\n"
                      "  // registers) and not a builtin the compiler may hoist above the begin marker\n"
                      "  asm volatile(\"v_mbcnt_lo_u32_b32 %0, -1, 0\\n\\tv_mbcnt_hi_u32_b32 %0, -1, %0\" : \"=v\"(ln));\n"
                      "  const bool t = (selmask >> ln) & 1;\n"
                      "  for (int k = 0; k < 8; k++) d.v[k] = t ? a.v[k] : b.v[k]; }",
                      vin=_fe(ra, "a") + _fe(rb, "b"), vout=_fe(D_REG, "d"), sin=[(S_SEL, "selmask")], parity=par))
        # checks: the status word lives in v124; sarg = index of the flat operation
        S.append(Body("asserteq_%s" % par, "if (!fe_eq(a, b)) cw_fail(st, CW_ST_ASSERT_FAILED, (uint32_t)sarg);",
                      vin=_fe(ra, "a") + _fe(rb, "b"), sin=[(S_ARG, "sarg")], keep_d=True, parity=par))
        S.append(Body("assertnz_%s" % par, "if (fe_is_zero(a)) cw_fail(st, CW_ST_ASSERT_FAILED, (uint32_t)sarg);",
                      vin=_fe(ra, "a"), sin=[(S_ARG, "sarg")], keep_d=True, parity=par))
        # fused R1CS check (fpjit.plan_checks): recomputed from the stored wires right after the row that produced the last
        # one; fb (v123) = smallest index of a violated constraint of this instance so far, sarg = the constraint's index
        S.append(Body("chkeq_%s" % par, "if (!fe_eq(a, b)) fb = min(fb, (uint32_t)sarg);",
                      vin=_fe(ra, "a") + _fe(rb, "b"), sin=[(S_ARG, "sarg")], keep_d=True, parity=par, chk=True))
        for nm, expr in (("chkmul", "fe_mmul(a, b, P)"), ("chkmul2", "fe_mul2_auto(a, b, P)"), ("chkadd", "fe_add(a, b, P)")):
            S.append(Body("%s_%s" % (nm, par), "if (!fe_eq(%s, g)) fb = min(fb, (uint32_t)sarg);" % expr,
                          vin=_fe(ra, "a") + _fe(rb, "b") + _fe(G_REG, "g"), sin=[(S_ARG, "sarg")], keep_d=True, parity=par, chk=True))
        # a linear row: g + reduce(acc) must equal the wire (or literal) in the A operand
        S.append(Body("chkdot_%s" % par, "{ const fe r = fe_add(g, fe_from29(fe29_reduce(acc, P)), P); for (int j = 0; j < 18; j++) acc[j] = 0;\n"
                      "  if (!fe_eq(r, a)) fb = min(fb, (uint32_t)sarg); }",
                      vin=_fe(ra, "a"), sin=[(S_ARG, "sarg")], keep_d=True, acc="dot", g=True, parity=par, chk=True))
        # LINSUM term: x in the A operand of the parity, |coefficient| in sarg; g / pos / neg accumulate across calls
        for sg, nm in ((0, "linp"), (1, "linn")):
            S.append(Body("%s_%s" % (nm, par), "linsum_term(a, sarg | %s, g, pos, neg, P);" % ("(1ull << 63)" if sg else "0ull"),
                          vin=_fe(ra, "a"), sin=[(S_ARG, "sarg")], keep_d=True, acc="lin", g=True, parity=par))
        # DOTC term: acc += x * (coef R') with the coefficient's nine 29-bit limbs in s[24:32]
        S.append(Body("dotmac_%s" % par, "{ uint32_t c29[9] = {c0, c1, c2, c3, c4, c5, c6, c7, c8}; fe29_mac(acc, fe_to29(a), c29); }",
                      vin=_fe(ra, "a"), sin=[(S_COEF + k, "c%d" % k) for k in range(9)], keep_d=True, acc="dot", parity=par))
    # HEAVY operators (thousands of instructions, many registers): operands always in the even set, nothing in flight across
    # them (the emitted code drains its prefetch first), so only the emitter's own registers are kept
    S.append(Body("inv_h", "d = fe_inv(a, P);", vin=_fe(A_E, "a"), vout=_fe(D_REG, "d"), parity="h"))
    S.append(Body("pow_h", "d = fe_pow(a, b, P);", vin=_fe(A_E, "a") + _fe(B_E, "b"), vout=_fe(D_REG, "d"), parity="h"))
    for nm, pick in (("idiv", "qq"), ("mod", "rr")):
        S.append(Body("%s_h" % nm,
                      "{ fe qq, rr; if (fe_is_zero(b)) { cw_fail(st, CW_ST_ARITH, (uint32_t)sarg); d = fe_zero(); }\n"
                      "  else { fe_divmod(a, b, &qq, &rr); d = %s; } }" % pick,
                      vin=_fe(A_E, "a") + _fe(B_E, "b"), vout=_fe(D_REG, "d"), sin=[(S_ARG, "sarg")], parity="h"))
    # D_CALL: a circom function with run-time control flow (tier 2): the interpreter of csrc/cw_call.hip.h as ONE body - the one
    # that works on memory (the call's register window in the value table, the bytecode, the constant table): sarg = function id
    # | first window slot << 32, c0 = index of the flat operation, c2:c3 = address of the kernel's argument block (the tables'
    # addresses are loaded from it: KA_* words), v120 / v121 = the lane's offsets.  Nothing else of the caller survives.
    S.append(Body("call_h", "{ const uint64_t *ka = (const uint64_t *)(((uint64_t)c3 << 32) | c2);\n"
                  "  EvalCtx c; c.Vb = (const char *)(((uint64_t)ks95 << 32) | ks94); c.Cb = (const char *)ka[%d];\n"
                  "  c.fcode = (const uint4 *)ka[%d]; c.ftab = (const uint4 *)ka[%d]; c.Lb = nullptr; c.terms = nullptr; c.tp = 0;\n"
                  "  c.vlo = vlo_; c.vhi = vhi_; c.lane16 = 0; c.lds_hi = 0; c.slot_stride = ks100;\n"
                  "  eval_call_body((uint32_t)sarg, (uint64_t)(uint32_t)(sarg >> 32) * (uint64_t)ks100, c0, st, c, P); }" % (KA_CONSTS, KA_FCODE, KA_FTAB),
                  vin=[(120, "vlo_"), (121, "vhi_")], sin=[(S_ARG, "sarg"), (S_COEF, "c0"), (S_COEF + 2, "c2"), (S_COEF + 3, "c3")], parity="c"))
    # The same interpreter for the STRAND kernels (16 waves per workgroup: 128 VGPRs per wave; the compiler spills into a private
    # segment of ~550 bytes per lane) WITH the native long_div: what config 5's verifier calls.  Compiled as a unit of its own
    # (build_bodies): CW_NO_NATIVE_LONG_DIV, which keeps call_h inside the reach of its branches, is not defined for it.
    S.append(Body("call_k", next(b_ for b_ in S if b_.name == "call_h").code,
                  vin=[(120, "vlo_"), (121, "vhi_")], sin=[(S_ARG, "sarg"), (S_COEF, "c0"), (S_COEF + 2, "c2"), (S_COEF + 3, "c3")], parity="k"))
    # row ends of the accumulating operators (no table operand, no parity)
    S.append(Body("linfin", "g = fe_add(g, acc192_to_fe(pos), P); d = fe_sub(g, acc192_to_fe(neg), P);",
                  vout=_fe(D_REG, "d"), acc="lin", g=True))
    # at most four products per reduction (column bound of the 64-bit accumulators): the emitter calls dotred after every
    # fourth term and at the end; g carries the running result
    S.append(Body("dotred", "g = fe_add(g, fe_from29(fe29_reduce(acc, P)), P); for (int j = 0; j < 18; j++) acc[j] = 0;",
                  keep_d=True, acc="dot", g=True))
    S.append(Body("dotfin", "d = fe_add(g, fe_from29(fe29_reduce(acc, P)), P); for (int j = 0; j < 18; j++) acc[j] = 0;",
                  vout=_fe(D_REG, "d"), acc="dot", g=True))
    # end of the strand: the first failed check of the instance reaches the status array (the one body with memory
    # instructions: nothing of the emitted code is in flight behind it)
    # (the findings of the fused R1CS check go to the second half of the status array - Bp words behind the first, Bp =
    # slot stride / 32 - where cw_check_r1cs picks them up)
    S.append(Body("publish", "{ uint32_t *status = (uint32_t *)(((uint64_t)ks97 << 32) | ks96);\n"
                  "  if (st && kv127 < ks93) cw_publish_status(status, kv127, st);\n"
                  "  if (fb != 0xFFFFFFFFu && kv127 < ks93) atomicMin(&status[(ks100 >> 5) + kv127], fb); }", keep_d=True, parity="m", chk=True))
    return S


def _source(bodies, fe_slow_inline=True, native_long_div=False):
    L = ['#include <hip/hip_runtime.h>', '#define CW_FE_SLOW __forceinline__' if fe_slow_inline else '',
         '#define CW_CALL_NATIVE __forceinline__', '' if native_long_div else '#define CW_NO_NATIVE_LONG_DIV 1', '#include "%s/cw_call.hip.h"' % CSRC]
    pins = _p_pins()
    for b in bodies:
        # heavy: the emitter re-derives its lane offsets afterwards; inv_h cannot even spare the status word's register
        # (the emitter parks it in LDS around the call)
        keep_v = set(V_OWNED) if b.parity not in ("h", "c", "k") else {124}
        no_st = b.name == "inv_h"
        if b.parity == "e":
            keep_v |= set(range(A_O, A_O + 16))
        elif b.parity == "o":
            keep_v |= set(range(A_E, A_E + 16))
        elif b.parity in (None, "m"):
            keep_v |= set(range(A_E, A_E + 16)) | set(range(A_O, A_O + 16))
        if b.keep_d:
            keep_v |= set(range(D_REG, D_REG + 8))
        vin = dict(b.vin)
        vout = dict(b.vout)
        st_reg = 124
        decl = ["fe a = fe_zero(), b = fe_zero(), d = fe_zero(), g = fe_zero();", "uint32_t st, vlo_ = 0, vhi_ = 0; uint64_t sarg = 0, selmask = 0;",
                "FpParams P;", "uint32_t c0, c1, c2, c3, c4, c5, c6, c7, c8;"]
        outs, ins = [], []         # of the BEGIN / END statements
        # vector inputs
        for r, e in vin.items():
            outs.append('"={v%d}"(%s)' % (r, e))
            keep_v.discard(r)
        # accumulators
        if b.acc == "dot":
            decl.append("uint64_t acc[18]; uint32_t al[18], ah[18];")
            for j in range(18):
                outs.append('"={v%d}"(al[%d])' % (ACC_REG + 2 * j, j))
                outs.append('"={v%d}"(ah[%d])' % (ACC_REG + 2 * j + 1, j))
                ins.append('"{v%d}"((uint32_t)acc[%d])' % (ACC_REG + 2 * j, j))
                ins.append('"{v%d}"((uint32_t)(acc[%d] >> 32))' % (ACC_REG + 2 * j + 1, j))
        elif b.acc == "lin":
            decl.append("Acc192 pos, neg; uint32_t al[12];")
            for j in range(12):
                outs.append('"={v%d}"(al[%d])' % (ACC_REG + j, j))
            for j, w in enumerate(("pos.w0", "pos.w1", "pos.w2", "neg.w0", "neg.w1", "neg.w2") if b.name != "linfin" else ()):
                ins.append('"{v%d}"((uint32_t)%s)' % (ACC_REG + 2 * j, w))
                ins.append('"{v%d}"((uint32_t)(%s >> 32))' % (ACC_REG + 2 * j + 1, w))
        if b.g:
            for k in range(8):
                outs.append('"={v%d}"(g.v[%d])' % (G_REG + k, k))
                if b.name not in ("linfin", "dotfin"):
                    ins.append('"{v%d}"(g.v[%d])' % (G_REG + k, k))
        elif b.acc:
            keep_v |= set(range(G_REG, G_REG + 8))
        # status word: read and written by the checks, kept otherwise
        if not no_st:
            outs.append('"={v%d}"(st)' % st_reg)
            ins.append('"{v%d}"(st)' % st_reg)
        keep_v.discard(st_reg)
        if b.chk:
            decl.append("uint32_t fb;")
            outs.append('"={v%d}"(fb)' % V_FB)
            ins.append('"{v%d}"(fb)' % V_FB)
            keep_v.discard(V_FB)
        for r in sorted(keep_v):
            if r in vout:
                continue
            decl.append("uint32_t kv%d;" % r)
            outs.append('"={v%d}"(kv%d)' % (r, r))
            ins.append('"{v%d}"(kv%d)' % (r, r))
        for r, e in vout.items():
            ins.append('"{v%d}"(%s)' % (r, e))
        # scalar inputs / outputs / kept
        decl.append("uint32_t sa_lo, sa_hi, sm_lo, sm_hi, kret0, kret1;")
        keep_s = {S_SEL, S_SEL + 1}
        sin = dict(b.sin)
        if S_ARG in sin:
            outs.append('"={s%d}"(sa_lo)' % S_ARG)
            outs.append('"={s%d}"(sa_hi)' % (S_ARG + 1))
        if S_SEL in sin:
            outs.append('"={s%d}"(sm_lo)' % S_SEL)
            outs.append('"={s%d}"(sm_hi)' % (S_SEL + 1))
        for r, e in sin.items():
            if r not in (S_ARG, S_SEL):
                outs.append('"={s%d}"(%s)' % (r, e))
        if dict(b.sout).get(S_SEL):
            ins.append('"{s%d}"((uint32_t)selmask)' % S_SEL)
            ins.append('"{s%d}"((uint32_t)(selmask >> 32))' % (S_SEL + 1))
        elif S_SEL in sin or True:
            if S_SEL not in sin:
                outs.append('"={s%d}"(sm_lo)' % S_SEL)
                outs.append('"={s%d}"(sm_hi)' % (S_SEL + 1))
            ins.append('"{s%d}"(sm_lo)' % S_SEL)
            ins.append('"{s%d}"(sm_hi)' % (S_SEL + 1))
        # the emitted code's own scalars, incl. the return address
        for r in range(S_OWNED, 102):
            decl.append("uint32_t ks%d;" % r)
            outs.append('"={s%d}"(ks%d)' % (r, r))
            ins.append('"{s%d}"(ks%d)' % (r, r))
        for r, e in pins:                  # the field parameters: defined at the start AND still there at the end
            outs.append('"={s%d}"(%s)' % (r, e))
            ins.append('"{s%d}"(%s)' % (r, e))
        # (the tier-2 interpreter only runs in single-strand programs: one wave per workgroup, up to 512 registers)
        L.append("__global__ void __launch_bounds__(%d) body_%s(uint32_t *sink) {" % (64 if b.parity == "c" else 1024, b.name))
        L.extend("  " + x for x in decl)
        L.append('  asm volatile("; BODY_BEGIN %s" : %s);' % (b.name, ", ".join(outs)))
        if S_ARG in sin:
            L.append("  sarg = ((uint64_t)sa_hi << 32) | sa_lo;")
        if S_SEL in sin:
            L.append("  selmask = ((uint64_t)sm_hi << 32) | sm_lo;")
        if b.acc == "dot":
            L.append("  for (int j = 0; j < 18; j++) acc[j] = ((uint64_t)ah[j] << 32) | al[j];")
        elif b.acc == "lin":
            L.append("  pos.w0 = ((uint64_t)al[1] << 32) | al[0]; pos.w1 = ((uint64_t)al[3] << 32) | al[2]; pos.w2 = ((uint64_t)al[5] << 32) | al[4];")
            L.append("  neg.w0 = ((uint64_t)al[7] << 32) | al[6]; neg.w1 = ((uint64_t)al[9] << 32) | al[8]; neg.w2 = ((uint64_t)al[11] << 32) | al[10];")
        L.append("  " + b.code)
        L.append('  asm volatile("; BODY_END %s" :: %s);' % (b.name, ", ".join(ins)))
        L.append("}")
    return "\n".join(L) + "\n"


RET_MARK = "  ; RETURN"
_CONST_MOV = re.compile(r"^\s*(v_mov_b32_e32|v_mov_b64_e32|s_mov_b32|s_mov_b64|s_movk_i32|s_brev_b32|v_bfrev_b32_e32)\s+[vs](\d+|\[\d+:\d+\]),\s*(-?\d+(\.\d+)?|0x[0-9a-fA-F]+)\s*$")
_SCRATCH = re.compile(r"^\s*scratch_")
_MEM = re.compile(r"^\s*(global_|flat_|buffer_|scratch_|ds_|s_load|s_buffer_load|s_store|s_swappc|s_call|s_setpc|s_getpc|s_endpgm|s_barrier|s_sendmsg|s_trap)")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_PGLABEL = re.compile(r"^(\.Lpost_getpc\d+):")
LONG_BRANCH = "  s_branch_long"          # pseudo instruction of a body's text: `s_branch_long <label> s[a:b]` (expand_long_branch)


def _collapse_long_branches(text):
    """The compiler relaxes a branch that cannot reach its target (+-128 KB) into
        s_getpc_b64 s[a:b] / .Lpost_getpcN: / s_add_u32 sa, sa, (T-.Lpost_getpcN)&4294967295 / s_addc_u32 sb, sb, (T-...)>>32 /
        s_setpc_b64 s[a:b]
    which is an unconditional branch to T: one pseudo line here (the checks treat it as s_branch), the five instructions again
    when a program prints the body (expand_long_branch)."""
    out, i = [], 0
    while i < len(text):
        t = text[i].strip()
        if t.startswith("s_getpc_b64") and i + 4 < len(text):
            pair = t.split()[1]
            lab = _PGLABEL.match(text[i + 1].strip())
            m1 = re.match(r"s_add_u32 s(\d+), s\1, \((\.LBB\d+_\d+)-(\.Lpost_getpc\d+)\)&4294967295$", text[i + 2].strip())
            m2 = re.match(r"s_addc_u32 s(\d+), s\1, \((\.LBB\d+_\d+)-(\.Lpost_getpc\d+)\)>>32$", text[i + 3].strip())
            m3 = re.match(r"s_setpc_b64 (s\[\d+:\d+\])$", text[i + 4].strip())
            if lab and m1 and m2 and m3 and m3.group(1) == pair and m1.group(2) == m2.group(2) and m1.group(3) == lab.group(1) == m2.group(3):
                out.append("%s %s %s" % (LONG_BRANCH, m1.group(2), pair))
                i += 5
                continue
        out.append(text[i])
        i += 1
    return out


_long_branch_serial = [0]


def expand_long_branch(t):
    """the instructions of a `s_branch_long` pseudo line (any other line: itself)"""
    if not t.startswith(LONG_BRANCH):
        return [t]
    _, lbl, pair = t.split()
    m = re.match(r"s\[(\d+):(\d+)\]", pair)
    a, b_ = m.group(1), m.group(2)
    _long_branch_serial[0] += 1
    pg = ".Lfj_pg%d" % _long_branch_serial[0]
    return ["  s_getpc_b64 %s" % pair, "%s:" % pg, "  s_add_u32 s%s, s%s, (%s-%s)&4294967295" % (a, a, lbl, pg),
            "  s_addc_u32 s%s, s%s, (%s-%s)>>32" % (b_, b_, lbl, pg), "  s_setpc_b64 %s" % pair]


def _regs(rx, text):
    out = set()
    for m in rx.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


_NO_DST = re.compile(r"^(s_cmp|s_cmpk|s_bitcmp|s_cbranch|s_branch|s_nop|s_waitcnt|s_setprio|s_sleep|s_barrier|scratch_store|global_store|"
                     r"flat_store|global_atomic|flat_atomic|v_nop|s_setpc)")
_TWO_DST = re.compile(r"^(v_mad_u64_u32|v_mad_i64_i32|v_add_co_u32|v_sub_co_u32|v_subrev_co_u32|v_addc_co_u32|v_subb_co_u32|"
                      r"v_subbrev_co_u32|v_div_scale|v_swap_b32)")


def _use_before_def(b, defined_v, defined_s):
    """A body is cut out of a kernel: anything the compiler computed BEFORE the begin marker (a constant it hoisted, the
    work-item id the hardware left in v0) is not there when the emitted code calls it.  May-define analysis over the
    body's basic blocks: reports a register that is read where NO path from the entry has written it and that is not one
    of the body's declared inputs."""
    # basic blocks
    blocks, cur, label_of = [], [], {}
    for t in b.text:
        t = t.strip()
        if t == RET_MARK.strip():
            cur.append("s_setpc_b64 ret")
            blocks.append(cur)
            cur = []
            continue
        if t.endswith(":"):
            if cur or not blocks:
                blocks.append(cur)
                cur = []
            label_of[t[:-1]] = len(blocks)
            continue
        cur.append(t)
        if t.startswith("s_branch") or t.startswith("s_cbranch"):      # (s_branch_long included)
            blocks.append(cur)
            cur = []
    blocks.append(cur)
    # a label directly behind a branch: the empty block appended above keeps indices consistent
    succ = []
    vfree = set()
    for k, blk in enumerate(blocks):
        last = blk[-1] if blk else ""
        if last.startswith("s_branch"):
            succ.append([label_of[last.split()[1]]])
        elif last.startswith("s_setpc"):
            succ.append([])
        elif last.startswith("s_cbranch"):
            succ.append([label_of[last.split()[1]]] + ([k + 1] if k + 1 < len(blocks) else []))
            if last.startswith("s_cbranch_execz"):     # taken with no lane active: what the vector registers hold is immaterial
                vfree.add((k, label_of[last.split()[1]]))
        else:
            succ.append([k + 1] if k + 1 < len(blocks) else [])
    pred = [[] for _ in blocks]
    for k, ss in enumerate(succ):
        for x in ss:
            pred[x].append(k)

    def key(kind, r):
        return r if kind == "v" else 1000 + r

    def walk(blk, live, report):
        for t in blk:
            parts = t.split(None, 1)
            op = parts[0]
            opnds = [x.strip() for x in parts[1].split(",")] if len(parts) > 1 else []
            nd = 0 if _NO_DST.match(op) else 2 if _TWO_DST.match(op) else 1
            if op == "s_branch_long":                  # `s_branch_long <label> s[a:b]`: writes the pair, reads nothing
                live |= {key("s", r) for r in _regs(_SREG, t.split()[2])}
                continue
            dsts, srcs = opnds[:nd], opnds[nd:]
            if op.startswith("v_swap"):
                srcs = opnds
            if report:
                for x in srcs:
                    for r in _regs(_VREG, x):
                        if key("v", r) not in live:
                            return "v%d in `%s`" % (r, t)
                    for r in _regs(_SREG, x):
                        if key("s", r) not in live:
                            return "s%d in `%s`" % (r, t)
            for x in dsts:
                live |= {key("v", r) for r in _regs(_VREG, x)} | {key("s", r) for r in _regs(_SREG, x)}
        return None

    entry = {key("v", r) for r in defined_v} | {key("s", r) for r in defined_s}
    allv = {key("v", r) for r in range(256)}
    universe = allv | {key("s", r) for r in range(110)}
    IN = [set() for _ in blocks]
    IN[0] = set(entry)
    OUT = [None] * len(blocks)
    changed = True
    while changed:
        changed = False
        for k, blk in enumerate(blocks):
            if k:
                ps = [(OUT[p_] | allv if (p_, k) in vfree else OUT[p_]) for p_ in pred[k] if OUT[p_] is not None]
                # MAY-define (union): the compiler's structurised control flow correlates its branches (a block that is skipped
                # implies another one ran), which a must-define intersection cannot see; a value hoisted above the marker
                # is defined on NO path, and that is what this check is for
                new_in = set.union(*ps) if ps else set(universe)
            else:
                new_in = set(entry)
            live = set(new_in)
            walk(blk, live, False)
            if OUT[k] != live or IN[k] != new_in:
                IN[k], OUT[k] = new_in, live
                changed = True
    for k, blk in enumerate(blocks):
        bad = walk(blk, set(IN[k]), True)
        if bad:
            return bad
    return None


def parse_bodies(asm: str, bodies):
    """cut every body out of the compiler's assembly and check it against its register budget"""
    by_name = {b.name: b for b in bodies}
    lines = asm.split("\n")
    i = 0
    while i < len(lines):
        m = re.search(r"; BODY_BEGIN (\w+)", lines[i])
        if not m:
            i += 1
            continue
        b = by_name[m.group(1)]
        j = i + 1
        # Constants the compiler materialised AHEAD of the begin marker (nothing orders a `v_mov_b32 v23, 0` behind an asm
        # statement it does not depend on) are part of the body: they are pulled in here; anything else the body would need
        # from the kernel's prologue is refused by the use-before-define check below
        text = []
        k0 = i
        while k0 > 0 and not lines[k0].startswith("_Z"):
            k0 -= 1
        for k2 in range(k0 + 1, i):
            ln = lines[k2].split(";")[0].rstrip() if not lines[k2].lstrip().startswith(";") else ""
            if _CONST_MOV.match(ln):
                text.append(ln)
        while "; BODY_END " + b.name not in lines[j]:
            ln = lines[j].split(";")[0].rstrip() if not lines[j].lstrip().startswith(";") else ""
            if ln.strip():
                text.append(ln)
            j += 1
        # Blocks the compiler placed BEHIND the end marker (cold paths moved to the end of the function: they are reached by
        # a branch from the body and branch back into it): everything between the kernel's s_endpgm and the end of the
        # function belongs to the body; it goes behind the body's return (RET_MARK is replaced by the emitter's s_setpc)
        jj = j
        while not lines[jj].startswith(".Lfunc_end"):
            jj += 1
        tail, seen_end = [], False
        for k2 in range(j, jj):
            raw = lines[k2]
            ln = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";") else ""
            if not seen_end:
                seen_end = ln.strip() == "s_endpgm"
                continue
            if ln.strip() and (not ln.strip().startswith(".") or _LABEL.match(ln.strip()) or _PGLABEL.match(ln.strip())):
                tail.append(ln)                 # an instruction or a block label; assembler directives are not code
        if tail:
            text = text + [RET_MARK] + tail
        # the compiler's own markers around the asm statements
        text = [t for t in text if "#ASMSTART" not in t and "#ASMEND" not in t]
        text = _collapse_long_branches(text)
        allowed_v = set(range(40, 120)) | {r for r, _ in b.vin} | {r for r, _ in b.vout} | {124} | ({V_FB} if b.chk else set())
        if b.parity == "e":
            allowed_v |= set(range(A_E, A_E + 16))
        elif b.parity == "o":
            allowed_v |= set(range(A_O, A_O + 16))
        elif b.parity == "h":
            allowed_v |= set(range(0, 128))
        elif b.parity == "c":
            allowed_v |= set(range(0, 256))
        elif b.parity == "k":
            allowed_v |= set(range(0, 128))
        if not b.keep_d:
            allowed_v |= set(range(D_REG, D_REG + 8))
        if b.acc and not b.g:
            allowed_v -= set(range(G_REG, G_REG + 8))
        allowed_s = set(range(0, 40)) | set(range(S_PARAMS, S_PARAMS + N_PARAM_SGPRS))
        if b.parity in ("m", "c", "k"):  # the strand's last call / the interpreter read the emitter's registers (status array, table)
            allowed_v |= set(V_OWNED)
            allowed_s |= set(range(S_OWNED, 102))
        out = []
        for t in text:
            if t == RET_MARK:
                out.append(t)
                continue
            if t.startswith(LONG_BRANCH):                 # (bodies beyond the 128 KB reach of s_branch: the interpreter)
                _, lbl, pair = t.split()
                us = _regs(_SREG, pair)
                if not us <= allowed_s:
                    raise RuntimeError("body %s: long branch through SGPRs outside its budget: %s" % (b.name, pair))
                out.append("%s %s %s" % (LONG_BRANCH, re.sub(r"\.LBB(\d+_\d+)", lambda mm: ".Lfj_%s_%s" % (b.name, mm.group(1)), lbl), pair))
                continue
            if b.parity in ("h", "c", "k") and _SCRATCH.match(t):
                b.scratch = True          # a heavy body may spill: the emitted kernel then owns a private segment
            elif b.parity in ("c", "k") and re.match(r"^\s*(global_|flat_|s_load|s_waitcnt)", t):
                pass                      # the tier-2 interpreter works on memory
            elif b.parity == "m" and re.match(r"^\s*(global_|flat_|s_waitcnt)", t):
                pass
            elif _MEM.match(t):
                raise RuntimeError("body %s contains an instruction a leaf body must not have: %s" % (b.name, t.strip()))
            lm = _LABEL.match(t.strip())
            if lm:
                out.append(".Lfj_%s_%s:" % (b.name, lm.group(1)[4:]))
                continue
            t = re.sub(r"\.LBB(\d+_\d+)", lambda mm: ".Lfj_%s_%s" % (b.name, mm.group(1)), t)
            uv = _regs(_VREG, t)
            if not uv <= allowed_v:
                raise RuntimeError("body %s touches VGPRs outside its budget: %s in `%s`" % (b.name, sorted(uv - allowed_v), t.strip()))
            us = _regs(_SREG, t)
            if not us <= allowed_s:
                raise RuntimeError("body %s touches SGPRs outside its budget: %s in `%s`" % (b.name, sorted(us - allowed_s), t.strip()))
            b.vwritten |= uv
            out.append(t)
        b.text = out
        b.n_agpr = max([int(x) + 1 for t in out for x in re.findall(r"\ba(\d+)\b", t)] +
                       [int(y) + 1 for t in out for _, y in re.findall(r"\ba\[(\d+):(\d+)\]", t)] + [0])
        # registers that hold something defined when the body is entered: its inputs and everything kept through it
        dv = {r for r, _ in b.vin} | set(V_OWNED) | {124}
        dv |= set(range(A_E, A_E + 16)) | set(range(A_O, A_O + 16))     # operand sets: own inputs, or kept for the other parity
        if b.keep_d or any(r == D_REG for r, _ in b.vin):
            dv |= set(range(D_REG, D_REG + 8))
        if b.g or b.acc:
            dv |= set(range(G_REG, G_REG + 8))
        if b.acc:
            dv |= set(range(ACC_REG, ACC_REG + (36 if b.acc == "dot" else 12)))
        dsr = set(range(S_PARAMS, S_PARAMS + N_PARAM_SGPRS)) | set(range(S_OWNED, 102)) | {S_SEL, S_SEL + 1}
        for r, _ in b.sin:
            dsr |= {r, r + 1} if r in (S_ARG, S_SEL) else {r}
        if b.parity == "k":
            # under 128 VGPRs the compiler saves every scalar that is live-in to its test kernel in VGPR lanes, s[0:1] (that
            # kernel's argument pointer, never used by the body) among them: a read of whatever the two hold, restored unused
            dsr |= {0, 1}
        bad = _use_before_def(b, dv, dsr)
        if bad:
            raise RuntimeError("body %s reads a register it neither receives nor writes first (hoisted above the marker?): %s" % (b.name, bad))
        if getattr(b, "scratch", False):
            m2 = None
            for k in range(j, len(lines)):
                m2 = re.search(r"; ScratchSize: (\d+)", lines[k])
                if m2:
                    break
            b.scratch_bytes = int(m2.group(1))
        b.n_instr = sum(4 if t.startswith(LONG_BRANCH) else 1 for t in out if not t.strip().endswith(":") and t != RET_MARK)
        i = j + 1
    missing = [b.name for b in bodies if b.text is None]
    if missing:
        raise RuntimeError("bodies not found in the compiler output: %s" % missing)
    for b in bodies:
        if b.parity not in ("h", "c", "k"):
            assert not getattr(b, "scratch", False)
    return bodies


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


_PARSED = {}


def build_bodies(names=None, cache=True):
    """name -> Body with its instruction text; compiled once per source (cached under circom_amd/lib)"""
    bodies = _specs()
    if names is not None:
        bodies = [b for b in bodies if b.name in names]
    units = [([b for b in bodies if b.parity != "k"], False), ([b for b in bodies if b.parity == "k"], True)]
    units = [(bs, nld) for bs, nld in units if bs]
    src = "\n// ---- unit ----\n".join(_source(bs, native_long_div=nld) for bs, nld in units)
    deps = b"".join(open(os.path.join(CSRC, n), "rb").read() for n in ("fp256.hip.h", "cw_rowops.hip.h", "cw_tape.h", "cw_call.hip.h"))
    key = hashlib.sha256(src.encode() + deps + open(__file__, "rb").read()).hexdigest()[:16]
    path = os.path.join(CACHE_DIR, "fpjit_bodies_%s.s" % key)
    # Cutting the bodies out of the text and checking their register use is 3 s of regular expressions: done once per process
    # (every emitted program of a session shares the dict: bodies are read-only after parsing) and once per source across
    # processes (the parsed bodies are pickled next to the assembly they came from)
    memo = (key, None if names is None else tuple(sorted(names)))
    if cache and memo in _PARSED:
        return _PARSED[memo]
    ppath = path[:-2] + (".pkl" if names is None else ".%s.pkl" % hashlib.sha256(repr(memo[1]).encode()).hexdigest()[:8])
    if cache and os.path.exists(ppath):
        try:
            with open(ppath, "rb") as f:
                got = pickle.load(f)
            if isinstance(got, dict) and all(getattr(b, "text", None) for b in got.values()):
                _PARSED[memo] = got
                return got
        except Exception:
            pass                                    # a torn or stale file: parse again below
    asm = None
    if cache and os.path.exists(path):
        asm = open(path).read()
    if asm is None:
        parts = []
        with tempfile.TemporaryDirectory(prefix="cw_fpjit_") as d:
            for ui, (bs, nld) in enumerate(units):
                s = os.path.join(d, "bodies%d.hip" % ui)
                with open(s, "w") as f:
                    f.write(_source(bs, native_long_div=nld))
                # the default (max-occupancy) machine scheduler of this LLVM crashes on some bodies under the physical-register
                # constraints; the max-ILP strategy does not, and ILP is what a row body wants anyway
                r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-mllvm",
                                    "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-disable-promote-alloca-to-lds", "-I" + CSRC, "-o",
                                    os.path.join(d, "bodies%d.s" % ui), s], capture_output=True, text=True)
                if r.returncode:
                    raise RuntimeError("hipcc failed on the row bodies:\n" + r.stderr[-4000:])
                parts.append(open(os.path.join(d, "bodies%d.s" % ui)).read())
        asm = "\n".join(parts)
        if cache:
            os.makedirs(CACHE_DIR, exist_ok=True)
            tmp = path + ".%d" % os.getpid()
            with open(tmp, "w") as f:
                f.write(asm)
            os.replace(tmp, path)
    parse_bodies(asm, bodies)
    got = {b.name: b for b in bodies}
    if cache:
        _PARSED[memo] = got
        try:
            tmp = ppath + ".%d" % os.getpid()
            with open(tmp, "wb") as f:
                pickle.dump(got, f, protocol=4)
            os.replace(tmp, ppath)
        except OSError:
            pass
    return got
