"""hip_elements bit-plane lowering, part 3: EMITTED gate code (the "sliced" engine, round 4).

The reference's back-end emits one C++ function per template (`compiler/src/circuit_design/template.rs:174-474`) and the
C++ compiler turns the witness program into machine code; the bit-plane engine of rounds 2-3 (bitsched.py + cw_bits.hip)
INTERPRETS the gate network instead - 15 instructions per 64 gates x 64 instances, every operand an LDS round trip, one
wave per SIMD because the LDS holds four groups per CU: 0.09 of the VALU issue roof and 0.11 of HBM.  This module is the
emitting counterpart for large batches: the gate network becomes STRAIGHT-LINE gfx950 code in which

  * a LANE holds 32 instances (one dword = the values of one signal in 32 instances), a wave 2 048 instances,
  * a GATE is one `v_bitop3_b32` on registers (any 3-input function, truth table in the instruction),
  * every distinct signal value is written ONCE to the bit table with a coalesced 256-byte `buffer_store_dword`
    (row = slot * 256 bytes inside the wave's CHUNK of the table: the witness image, 1 bit per signal and instance),
  * the R1CS check is fused: every non-trivial constraint becomes gates on the very registers that hold its wires (LUT
    class: the violation table of `A*B - C` over <= 6 wires; integer class: both sides of a long linear row summed by
    carry-save adders and compared bit by bit) OR-ed into one flag word per lane; `cw_check_r1cs` then only has to
    audit groups whose flag is set (the stand-alone check kernels stay as the independent audit),
  * assertion gates (unproved `===`) are OR-ed into the fallback mask as before.

Register allocation is done here (no compiler sees the code): 253 VGPRs with furthest-next-use eviction, 256 AccVGPRs as
the second level (`v_accvgpr_write/read`: one instruction each way, no latency), the bit table itself as the third
(every signal value has a row anyway; other values get scratch rows).  Values that must come back from memory are
requested PREFETCH instructions ahead; `s_waitcnt vmcnt(N)` immediates are computed exactly from the in-order issue
count.  `tools/ubench_icache.hip` (profiles/r04_ubench_icache.*) is the measurement behind the design: a wave sustains
one instruction per 4.4-5.0 clocks on 8 MB of straight-line code (64 KB instruction cache), two waves per SIMD twice that.

The IR this module produces is executed by `oracle/jit_eval.py` on the CPU (poisoned registers, in-order memory queue:
a wrong wait count or a clobbered register raises) before any GPU sees it; `to_asm` prints it as gfx950 assembly and
`assemble` runs the LLVM assembler + linker of the ROCm installation (no hipcc: there is nothing to compile).
"""
from __future__ import annotations

import heapq
import os
import subprocess
import tempfile

import numpy as np

IN_BASE = 3                   # slot of main input 0 (0 = constant 0, 1 = constant ones, 2 = reserved): cw_bits_host.h
ROW_BYTES = 256               # one slot of one chunk: 64 lanes x 4 bytes = 2 048 instances
CHUNK_INSTANCES = 2048
PAGE = 4096                   # reach of the 12-bit immediate offset of a MUBUF instruction: 16 rows
N_PAGE_SGPRS = 12
KERNEL_NAME = "cw_bits_jit"
INF = 1 << 60

# v_bitop3_b32 index = s0 << 2 | s1 << 1 | s2; BitNet tables index = a | b << 1 | c << 2  ->  sources are (c, b, a)
TT_XOR3, TT_MAJ, TT_XOR2, TT_AND2 = 0x96, 0xE8, 0x66, 0x88


def _llvm_bin():
    for d in (os.environ.get("CW_LLVM_BIN"), "/opt/rocm/lib/llvm/bin", "/opt/rocm/llvm/bin"):
        if d and os.path.exists(os.path.join(d, "clang")):
            return d
    raise RuntimeError("no ROCm LLVM tools found (clang / ld.lld): set CW_LLVM_BIN")


# ---- the check network ----------------------------------------------------------------------------------------------------
class _Gates:
    """3-input gates appended behind the evaluation network's nodes, with their own structural sharing (nothing is shared
    with the evaluation's gates: the check recomputes from the WIRES, i.e. from the stored signal values)."""

    def __init__(self, first_id):
        self.first = first_id
        self.tt, self.a, self.b, self.c = [], [], [], []
        self.rowkey = []                  # per gate: completion key of the constraint that created it (lower_jit orders by it)
        self.cur_key = 0
        self.cse = {}

    def gate(self, tt, ops):
        """function `tt` (index = x0 | x1 << 1 | x2 << 2) of up to three operand nodes; constants (0 / 1) and repeated
        operands are folded away"""
        ops = list(ops) + [0] * (3 - len(ops))
        # fold constants / duplicates: evaluate over the distinct variable operands
        var = []
        for o in ops:
            if o > 1 and o not in var:
                var.append(o)
        var.sort()
        n = len(var)
        t = 0
        for m in range(1 << n):
            idx = 0
            for j, o in enumerate(ops):
                bit = o if o <= 1 else (m >> var.index(o)) & 1
                idx |= bit << j
            t |= ((tt >> idx) & 1) << m
        # drop variables the function does not depend on
        j = 0
        while j < n:
            lo = hi = 0
            k = 0
            for m in range(1 << n):
                if not (m >> j) & 1:
                    lo |= ((t >> m) & 1) << k
                    hi |= ((t >> (m | (1 << j))) & 1) << k
                    k += 1
            if lo == hi:
                var.pop(j)
                t = lo
                n -= 1
            else:
                j += 1
        if n == 0:
            return t & 1
        if n == 1 and t == 2:
            return var[0]
        if n == 1:
            t8 = 0xAA if t == 2 else 0x55
        elif n == 2:
            t8 = t | (t << 4)
        else:
            t8 = t
        key = (t8,) + tuple(var + [0] * (3 - n))
        nid = self.cse.get(key)
        if nid is None:
            nid = self.first + len(self.tt)
            self.tt.append(t8)
            self.a.append(key[1]); self.b.append(key[2]); self.c.append(key[3])
            self.rowkey.append(self.cur_key)
            self.cse[key] = nid
        return nid

    def lut(self, leaves, tt):
        """any function of `leaves` (table index bit j = leaves[j]) as a small network of 3-input gates: the recipe is found
        once per distinct table (`lut_recipe`: a circuit has a handful - Sha256: 16) and replayed on the leaves"""
        steps = lut_recipe(len(leaves), tt)
        val = []
        for t8, ops in steps:
            val.append(self.gate(t8, [leaves[o] if o >= 0 else val[-1 - o] for o in ops]))
        return val[-1] if steps else (tt & 1)

    def sum_bits(self, terms):
        """bits (LSB first, node ids) of sum(coef * node), coef > 0: carry-save columns, oldest entries first"""
        cols = {}
        for node, coef in terms:
            k = 0
            while coef:
                if coef & 1:
                    cols.setdefault(k, []).append(node)
                coef >>= 1
                k += 1
        out = []
        k = 0
        while cols:
            col = cols.pop(k, [])
            i = 0
            while len(col) - i > 1:
                if len(col) - i >= 3:
                    x, y, z = col[i], col[i + 1], col[i + 2]
                    i += 3
                    s, cy = self.gate(TT_XOR3, (x, y, z)), self.gate(TT_MAJ, (x, y, z))
                else:
                    x, y = col[i], col[i + 1]
                    i += 2
                    s, cy = self.gate(TT_XOR2, (x, y)), self.gate(TT_AND2, (x, y))
                if s:
                    col.append(s)
                if cy:
                    cols.setdefault(k + 1, []).append(cy)
            out.append(col[i] if len(col) > i else 0)
            k += 1
        return out


# ---- 3-input-gate networks for functions of up to 6 variables -------------------------------------------------------------------
# A recipe is a list of steps (table over its <= 3 operands, operands); an operand >= 0 is a variable, -1 - k is step k.
# Search: functional decomposition on every bound set of three variables (column multiplicity 2: one gate feeds a gate over
# the free variables; multiplicity <= 4: two gates encode the column class, every class encoding is tried) and Shannon
# expansion on every variable, recursively on the residual function, cheapest first; memoised per (n, table).
_RECIPES = {}


def _cofactors(n, tt, j):
    lo = hi = 0
    k = 0
    for m in range(1 << n):
        if not (m >> j) & 1:
            lo |= ((tt >> m) & 1) << k
            hi |= ((tt >> (m | (1 << j))) & 1) << k
            k += 1
    return lo, hi


def _shift_vars(steps, vmap, base):
    """re-address a recipe: variable v -> vmap[v] (a variable >= 0 or a step reference < 0), its own steps start at `base`"""
    out = []
    for t8, ops in steps:
        out.append((t8, tuple(vmap[o] if o >= 0 else -1 - (base + (-1 - o)) for o in ops)))
    return out


def lut_recipe(n, tt):
    key = (n, tt)
    r = _RECIPES.get(key)
    if r is not None:
        return r
    full = (1 << (1 << n)) - 1
    tt &= full
    # support reduction
    for j in range(n):
        lo, hi = _cofactors(n, tt, j)
        if lo == hi:
            sub = lut_recipe(n - 1, lo)
            vmap = [v if v < j else v + 1 for v in range(n - 1)]
            r = _shift_vars(sub, vmap, 0)
            _RECIPES[key] = r
            return r
    if n == 0:
        r = []
    elif n <= 3:
        t8 = tt if n == 3 else (tt | (tt << 4)) if n == 2 else (0xAA if tt == 2 else 0x55)
        r = [(t8 & 0xFF, tuple(range(n)))]
    else:
        import itertools
        best = None

        def consider(c):
            nonlocal best
            if c is not None and (best is None or len(c) < len(best)):
                best = c

        # Shannon expansion on every variable: mux(x_j, f1, f0)
        for j in range(n):
            lo, hi = _cofactors(n, tt, j)
            vmap = [v if v < j else v + 1 for v in range(n - 1)]
            r0 = _shift_vars(lut_recipe(n - 1, lo), vmap, 0)
            r1 = _shift_vars(lut_recipe(n - 1, hi), vmap, len(r0))
            # a cofactor that is a constant or a plain variable has an empty recipe
            def ref(rec, f, off):
                if rec:
                    return -1 - (off + len(rec) - 1), None
                # f over n-1 vars after reduction is constant or a single variable / its complement handled as a gate above;
                # empty recipe = constant
                return None, f & 1
            a_ref, a_c = ref(r0, lo, 0)
            b_ref, b_c = ref(r1, hi, len(r0))
            steps = r0 + r1
            ops = []
            t = 0
            # final gate over (f0, f1, x_j) with constants folded in
            srcs = [(a_ref, a_c), (b_ref, b_c), (j, None)]
            var_ops = [s_[0] for s_ in srcs if s_[0] is not None]
            for m in range(1 << len(var_ops)):
                vals = []
                k = 0
                for rf, cst in srcs:
                    if rf is None:
                        vals.append(cst)
                    else:
                        vals.append((m >> k) & 1)
                        k += 1
                f0v, f1v, sv = vals
                t |= (f1v if sv else f0v) << m
            nv = len(var_ops)
            t8 = t if nv == 3 else (t | (t << 4)) if nv == 2 else (0xAA if t == 2 else 0x55)
            consider(steps + [(t8 & 0xFF, tuple(var_ops))])
        # decomposition on a bound set of three variables
        for T in itertools.combinations(range(n), 3):
            R = [v for v in range(n) if v not in T]
            nr = len(R)
            cols = []
            for t in range(8):
                col = 0
                for rr in range(1 << nr):
                    m = 0
                    for i, v in enumerate(T):
                        m |= ((t >> i) & 1) << v
                    for i, v in enumerate(R):
                        m |= ((rr >> i) & 1) << v
                    col |= ((tt >> m) & 1) << rr
                cols.append(col)
            distinct = sorted(set(cols))
            mu = len(distinct)
            if mu == 2:
                h = sum((1 << t) for t in range(8) if cols[t] == distinct[1])
                # g over (h, R...): index bit 0 = h
                g = 0
                for m in range(1 << (1 + nr)):
                    g |= ((distinct[m & 1] >> (m >> 1)) & 1) << m
                sub = lut_recipe(1 + nr, g)
                vmap = [-1] + R                            # variable 0 of g = step 0 (h)
                consider([(h, T)] + _shift_vars(sub, vmap, 1))
            elif mu <= 4 and nr <= 2:
                for perm in itertools.permutations(range(4), mu):
                    code = {d: perm[i] for i, d in enumerate(distinct)}
                    h1 = sum((1 << t) for t in range(8) if code[cols[t]] & 1)
                    h2 = sum((1 << t) for t in range(8) if code[cols[t]] & 2)
                    if h1 in (0, 255) or h2 in (0, 255):
                        continue
                    used = {c: d for d, c in code.items()}
                    fillers = [None] if mu == 4 else distinct
                    for fill in fillers:
                        g = 0
                        for m in range(1 << (2 + nr)):
                            col = used.get(m & 3, fill)
                            g |= ((col >> (m >> 2)) & 1) << m
                        sub = lut_recipe(2 + nr, g)
                        vmap = [-1, -2] + R
                        consider([(h1, T), (h2, T)] + _shift_vars(sub, vmap, 2))
        r = best
    _RECIPES[key] = r
    return r


def _recipe_eval(n, steps, m):
    """value of a recipe under assignment m (tests)"""
    val = []
    for t8, ops in steps:
        idx = 0
        for k, o in enumerate(ops):
            idx |= (((m >> o) & 1) if o >= 0 else val[-1 - o]) << k
        val.append((t8 >> idx) & 1)
    return val[-1]


LUT_MAX_WIRES = 6
INT_COEF_LIMIT = 1 << 64
SUBSTITUTE_PRODUCTS = True      # build_check: a row's table drops a wire that a checked product row `w = x * y` pins (x, y in the row too)


def _resolve_ports(net) -> dict:
    """port node -> the real node (gate, main input, constant 0 / 1) whose value it carries, chains followed"""
    src = getattr(net, "port_src", None) or {}
    out = {}
    for p_ in src:
        x = p_
        while x in src:
            x = src[x]
        out[p_] = x
    return out


def build_check(net, fc):
    """Fused R1CS check: gates behind the evaluation network whose leaves are the nodes of the constraint wires.
    Returns (_Gates, viol nodes, stats).  stats['unchecked'] = constraints the fused check does not cover (the stand-alone
    audit kernels then stay mandatory for the circuit)."""
    q = fc.fp.q
    half = q >> 1
    sn = net.sig_node
    G = _Gates(len(net.tt))
    viol = []
    st = {"trivial": 0, "lut": 0, "int": 0, "unchecked": 0}
    seen = {}

    def signed(v):
        v %= q
        return v - q if v > half else v

    ports = _resolve_ports(net)

    def lin(d, through_ports=False):
        """{signal: coef} -> (c0, {node: coef}) in signed representatives"""
        c0 = 0
        t = {}
        for s, cf in d.items():
            nd = int(sn[s]) if s else 1
            if through_ports:
                nd = ports.get(nd, nd)
            if nd == 0:
                continue
            cf = signed(cf)
            if nd == 1:
                c0 += cf
            else:
                t[nd] = t.get(nd, 0) + cf
        return signed(c0), {k: signed(v) for k, v in t.items() if v % q}

    # Wires that a CHECKED product row pins: `w = x * y` (single terms whose coefficients cancel) over three distinct wires (Xor3's and Maj's
    # `mid <== b * c`).  Another row of up to LUT_MAX_WIRES wires that contains w, x and y is checked with w REPLACED by x & y -
    # one wire less in its table (Xor3's `out` row: 5 -> 4 wires, 4 -> 2 gates) and no need to keep w around for it.  The
    # instance-level verdict is unchanged: if the product row holds, w = x & y and the other row is evaluated on the very same
    # values; if it does not hold, the product row itself raises the flag.  (Which row failed first comes from the audit
    # kernels, which check every row as written.)
    pinned = {}
    if SUBSTITUTE_PRODUCTS:
        for (A, B, C) in fc.constraints:
            if len(A) == 1 and len(B) == 1 and len(C) == 1:
                (sa, ca), (sb, cb), (sc, cc) = next(iter(A.items())), next(iter(B.items())), next(iter(C.items()))
                if sa and sb and sc and cc % q and (ca * cb - cc) % q == 0:          # ca x * cb y = cc w with ca cb = cc: w = x y
                    x, y, w = int(sn[sa]), int(sn[sb]), int(sn[sc])
                    if min(x, y, w) > 1 and len({x, y, w}) == 3 and w not in pinned and w not in ports:
                        pinned[w] = (x, y)
    st["substituted"] = 0

    for (A, B, C) in fc.constraints:
        a0, at = lin(A)
        b0, bt = lin(B)
        c0, ct = lin(C)
        wires = sorted(set(at) | set(bt) | set(ct))
        # the gates of a constraint are placed where its YOUNGEST wire is produced (evaluation nodes are numbered in program
        # order): the whole row is then checked in one block, on registers the evaluation has just used
        G.cur_key = wires[-1] if wires else 0
        if not wires:
            if (a0 * b0 - c0) % q:
                st["unchecked"] += 1          # a constant constraint that does not hold: leave it to the audit kernels
            else:
                st["trivial"] += 1
            continue
        key = (a0, tuple(sorted(at.items())), b0, tuple(sorted(bt.items())), c0, tuple(sorted(ct.items())))
        if key in seen:
            st["trivial"] += 1                # the same relation over the same nodes was checked already
            continue
        seen[key] = True
        if ports and any(w in ports for w in wires):
            # a wire that is a PORT (the input signal of a repeated template's instance) carries its source's value by
            # construction - the emitted code reads it from the source's row: `hin[k] === previous out[k]`, which the parent's
            # `<==` adds, holds identically.  Such a row is trivial if it is with every port replaced by what it carries.
            ra0, rat = lin(A, True)
            rb0, rbt = lin(B, True)
            rc0, rct = lin(C, True)
            rw = sorted(set(rat) | set(rbt) | set(rct))
            if len(rw) <= LUT_MAX_WIRES:
                holds = True
                for m in range(1 << len(rw)):
                    av, bv, cv = ra0, rb0, rc0
                    for j, w in enumerate(rw):
                        if (m >> j) & 1:
                            av += rat.get(w, 0); bv += rbt.get(w, 0); cv += rct.get(w, 0)
                    if (av * bv - cv) % q:
                        holds = False
                        break
                if holds:
                    st["trivial"] += 1
                    continue
        if len(wires) <= LUT_MAX_WIRES:
            # wires of this row that a product row pins to two other wires of this row (not the product row itself)
            sub = {}
            if pinned:
                for w in wires:
                    xy = pinned.get(w)
                    if xy is not None and xy[0] in wires and xy[1] in wires and len(wires) > 3 and xy[0] not in sub and xy[1] not in sub:
                        sub[w] = xy
            free = [w for w in wires if w not in sub]
            n = len(free)
            tt = 0
            for m in range(1 << n):
                val = {w: (m >> j) & 1 for j, w in enumerate(free)}
                for w, (x, y) in sub.items():
                    val[w] = val[x] & val[y]
                av, bv, cv = a0, b0, c0
                for w in wires:
                    if val[w]:
                        av += at.get(w, 0); bv += bt.get(w, 0); cv += ct.get(w, 0)
                if (av * bv - cv) % q:
                    tt |= 1 << m
            if tt == 0:
                st["trivial"] += 1
                continue
            if sub:
                st["substituted"] += 1
            v = G.lut(free, tt)
            st["lut"] += 1
            if v:
                viol.append(v)
            continue
        # long rows: linear (one factor constant) with integer coefficients -> exact integer comparison of the two sides
        if at and bt:
            st["unchecked"] += 1
            continue
        # A*B - C = 0 with A constant (or B constant): k * (b0 + sum bt) - (c0 + sum ct) = 0
        if at:
            kf, k0, kt = b0, a0, at
        else:
            kf, k0, kt = a0, b0, bt
        tot0 = kf * k0 - c0
        tot = {}
        for w, cf in kt.items():
            tot[w] = tot.get(w, 0) + kf * cf
        for w, cf in ct.items():
            tot[w] = tot.get(w, 0) - cf
        tot = {w: signed(cf) for w, cf in tot.items()}
        tot0 = signed(tot0)
        mag = abs(tot0) + sum(abs(v) for v in tot.values())
        if mag >= half or any(abs(v) >= INT_COEF_LIMIT for v in tot.values()) or abs(tot0) >= INT_COEF_LIMIT:
            st["unchecked"] += 1
            continue
        pos = [(w, cf) for w, cf in sorted(tot.items()) if cf > 0]
        neg = [(w, -cf) for w, cf in sorted(tot.items()) if cf < 0]
        if tot0 > 0:
            pos.append((1, tot0))
        elif tot0 < 0:
            neg.append((1, -tot0))
        pb, nb = G.sum_bits(pos), G.sum_bits(neg)
        acc = 0
        for k in range(max(len(pb), len(nb))):
            x = pb[k] if k < len(pb) else 0
            y = nb[k] if k < len(nb) else 0
            acc = G.gate(0xF6, (x, y, acc))              # acc | (x ^ y): index = x | y << 1 | acc << 2
        st["int"] += 1
        if acc:
            viol.append(acc)
    return G, viol, st


# ---- allocation + emission ----------------------------------------------------------------------------------------------------
class JitProgram:
    def __init__(self):
        self.ir = None                  # list of tuples (see _Alloc.emit_*)
        self.n_slots = 0                # rows per chunk (inputs, signal values, scratch)
        self.sig_slot = None            # uint32 [n_signals]
        self.n_signals = 0
        self.n_inputs = 0
        self.n_vgpr = 256
        self.n_agpr = 256
        self.check_complete = False
        self.stats = {}
        self.code = None                # code object (ELF) bytes once assembled
        self.node_slot = None           # evaluation node -> row of the table (-1: none)
        self.audit_code = None          # code object of the stand-alone audit of this program's table (lower_jit(audit_of=))
        self.is_audit = False
        self.loop = None                # {"base", "R", "used", "first", "K"}: the rows of the loop's iterations (lower_jit(loop=True))


AUDIT_SCRATCH_ROWS = 128        # spare rows at the end of every iteration of a loop: scratch of the audit program's iteration


class _AuditTooBig(Exception):
    """the audit's iteration needs more scratch rows than the main program left per iteration"""


OPAQUE_MIN_SIGNALS = 20000     # instances of a repeated template at least this large: loop candidates (their inputs become ports)
LOOP_MIN_BODY_GATES = 2048


def loop_templates(fc, min_signals=None):
    """The reference emits ONE body per template (`<T>_<id>_run`, compiler/src/circuit_design/template.rs:160-474) and runs it
    once per instance; the candidates for that treatment here: templates instantiated at least twice with the same parameters
    whose instance (with its sub-components) has at least `min_signals` signals.  -> [(template instance id, [component
    indices in code order])], largest coverage first."""
    if not hasattr(fc, "comp_code_range") or getattr(fc, "prog", None) is None:
        return []
    if min_signals is None:
        min_signals = OPAQUE_MIN_SIGNALS
    insts = fc.prog.inst_list
    groups = {}
    for ci, iid in enumerate(fc.comp_inst):
        if ci and insts[iid].n_total >= min_signals:
            groups.setdefault(iid, []).append(ci)
    out = [(iid, sorted(cis, key=lambda c_: fc.comp_code_range[c_][0])) for iid, cis in groups.items() if len(cis) >= 2]
    out.sort(key=lambda e: -insts[e[0]].n_total * len(e[1]))
    return out


def instance_ports(fc, min_signals=None):
    """(ports, marks) for `bitblast(fc, ports=, marks=)`: per instance of a loop candidate the ranges of its INPUT signals -
    whatever the parent wires to them (the IV constants of the first SHA-256 block, the previous block's output, the padding of
    the last) they are run-time values of the reference's template body - and the rows of the flat code where each
    instance's code begins and ends.  (None, None) when the circuit has no candidate."""
    cands = loop_templates(fc, min_signals)
    if not cands:
        return None, None
    from ..frontend.dsl import _prod
    ports, marks = [], set()
    insts = fc.prog.inst_list
    for iid, cis in cands:
        for ci in cis:
            base = fc.comp_sigstart[ci]
            ports.append(sorted((base + off, _prod(dims)) for _, dims, off in insts[iid].decls["i"]))
            marks.update(fc.comp_code_range[ci])
    return ports, marks


def _plan_segments(net, fc, n_eval, TT, A, B, C, is_gate, live, G, flags, hoist, want_loop, al, main_loop=None, node_slot=None):
    """Order of evaluation, segment by segment.  Without a loop: ONE segment (the whole program).  With the instances of a
    repeated template found in the network (their node ranges: net.marks at FlatCircuit.comp_code_range): prologue | one segment
    per instance | epilogue, every gate ordered inside its own segment; the longest run of instances whose ordered gate lists
    are ISOMORPHIC (same tables, same operand structure, same stored / asserted / violation flags; operands from outside the
    instance - its ports, in general any value of another segment - form the same pattern) becomes the loop.
    -> (order: int64 array of gate ids, seg_of: int32 array over nodes, bounds: [start of segment i in order], loop) with
    loop = None | dict(first=segment index, K=iterations, ext=[K][n_ext] node ids)"""
    n_nodes = len(TT)
    An, Bn, Cn = (np.asarray(x, dtype=np.int64) for x in (A, B, C))
    gate_mask = np.asarray(is_gate, dtype=bool)
    live_mask = np.asarray(live, dtype=bool) & gate_mask
    cands = []
    if want_loop and getattr(net, "marks", None):
        for iid, cis in loop_templates(fc):
            rng = [(net.marks[r0], net.marks[r1]) for r0, r1 in (fc.comp_code_range[ci] for ci in cis) if r0 in net.marks and r1 in net.marks]
            if len(rng) >= 2 and all(rng[i][1] <= rng[i + 1][0] for i in range(len(rng) - 1)):
                cands.append(rng)
    rowkey = G.rowkey
    dbg = bool(os.environ.get("CW_JIT_LOOP_DEBUG"))
    for rng in cands + [None]:
        seg = np.zeros(n_nodes, dtype=np.int32)
        floors = [0]
        if rng is not None:
            ok = True
            for k, (n0, n1) in enumerate(rng):
                seg[n0:n1] = k + 1
                floors.append(n0)
                if k and live_mask[rng[k - 1][1]:n0].any():
                    ok = False                         # gates between two instances (glue of the parent): no clean iteration
            if not ok:
                if dbg:
                    print("loop candidate dropped: live gates between two instances")
                continue
            seg[rng[-1][1]:n_eval] = len(rng) + 1
            floors.append(rng[-1][1])
            seg[~gate_mask] = 0
            # the PORTS of an instance (created in front of its gates) belong to it: a constraint over ports alone is checked in
            # their instance's iteration, where the rows they are read from exist
            port_ids = np.fromiter(getattr(net, "port_src", {}).keys(), dtype=np.int64)
            if len(port_ids):
                lo = 0
                for k, (n0, n1) in enumerate(rng):
                    seg[port_ids[(port_ids >= lo) & (port_ids < n1)]] = k + 1      # (its own, and those of its sub-components)
                    lo = n1
        sl = seg.tolist()
        # check gates: the latest segment among their operands (wires that are inputs / ports / prologue values
        # do not pull a constraint out of the prologue)
        for nid in range(n_eval, n_nodes):
            sl[nid] = max(sl[al[A[nid]]], sl[al[B[nid]]], sl[al[C[nid]]]) if rng is None else max(sl[A[nid]], sl[B[nid]], sl[C[nid]])
        # keys: evaluation gates in creation order - except that a gate whose operands (of its own segment) were ALL created long
        # before it moves up behind the youngest of them (circomlib's SHA-256 computes a block twice: the `<--` hint function
        # first, then the constrained components, whose values are the hint's gates again plus a few of their own - the `mid`
        # products, partial sums: created a whole block later than everything they read and are read with); every check gate sits
        # where the youngest wire of its constraint is produced.  Operands of other segments count as "there from the start".
        kl = [0] * n_nodes
        for nid in range(2, n_eval):
            if is_gate[nid]:
                sg = sl[nid]
                m = -1
                km = floors[sg]
                for o in (A[nid], B[nid], C[nid]):
                    o = al[o]                      # (a port waits for the value it carries)
                    if o > 1 and is_gate[o] and sl[o] == sg:
                        if o > m:
                            m = o
                        if kl[o] > km:
                            km = kl[o]
                kl[nid] = nid if (m >= 0 and nid - m <= hoist) else km
        for nid in range(n_eval, n_nodes):
            sg = sl[nid]
            km = floors[sg]
            for o in (rowkey[nid - n_eval], A[nid], B[nid], C[nid]):
                o = al[o]
                if o > 1 and sl[o] == sg and kl[o] > km:
                    km = kl[o]
            kl[nid] = km
        seg = np.asarray(sl, dtype=np.int32)
        key = np.asarray(kl, dtype=np.int64)
        gates = np.nonzero(live_mask)[0]
        if flags["audit"]:
            gates = gates[gates >= n_eval]
        if len(gates) == 0:
            return None
        order = gates[np.lexsort((gates, key[gates], seg[gates]))]
        n_seg = len(floors)
        bounds = np.searchsorted(seg[order], np.arange(n_seg + 1)).tolist()
        if rng is None:
            return order, seg, bounds, None
        # ---- isomorphism of the instances -----------------------------------------------------------------------------
        import hashlib
        posn = np.zeros(n_nodes, dtype=np.int64)
        posn[order] = np.arange(len(order)) - np.asarray(bounds, dtype=np.int64)[seg[order]]
        TTn = np.asarray(TT, dtype=np.int64)
        fl = flags["signal"].astype(np.int64) | (flags["assert"].astype(np.int64) << 1) | (flags["viol"].astype(np.int64) << 2)
        digests, exts = [], []
        for k in range(len(rng)):
            g = order[bounds[k + 1]:bounds[k + 2]]
            ops = np.stack([An[g], Bn[g], Cn[g]], axis=1)                       # [L][3]
            internal = gate_mask[ops] & (seg[ops] == k + 1)
            code = np.where(internal, posn[ops], np.where(ops <= 1, -1 - ops, np.int64(-3)))
            if flags["audit"]:
                # the audit loads its wires: an evaluation gate of the same instance is a ROW of the main program's iteration
                # (the same local row in every iteration), only the check's own gates are positions of this program
                wire = internal & (ops < n_eval)
                it = k + 1 - main_loop["first"]
                if 0 <= it < main_loop["K"]:
                    code = np.where(wire, (np.int64(1) << 40) + node_slot[np.minimum(ops, n_eval - 1)] - (main_loop["base"] + it * main_loop["R"]), code)
                    internal = internal | wire               # (not an external value)
                else:
                    internal = internal & ~wire              # an instance outside the main program's loop: plain rows
            extm = (~internal) & (ops > 1)
            flat = ops[extm]
            if len(flat):
                uq, first, inv = np.unique(flat, return_index=True, return_inverse=True)
                rank = np.empty(len(uq), dtype=np.int64)
                by_first = np.argsort(first, kind="stable")
                rank[by_first] = np.arange(len(uq))
                code[extm] = -3 - rank[inv]
                exts.append(uq[by_first])
            else:
                exts.append(np.zeros(0, dtype=np.int64))
            h = hashlib.blake2b(digest_size=16)
            h.update(TTn[g].tobytes()); h.update(code.tobytes()); h.update(fl[g].tobytes())
            digests.append(h.digest())
            if dbg:
                if k == 0:
                    ref_ = (TTn[g], code, fl[g], g)
                else:
                    n_ = min(len(g), len(ref_[0]))
                    g, code = g[:n_], code[:n_]
                    ref_c = tuple(x[:n_] for x in ref_)
                    d_tt, d_code, d_fl = (TTn[g] != ref_c[0]), (code != ref_c[1]).any(axis=1), (fl[g] != ref_c[2])
                    print("  instance %d vs 0: tables differ at %d, operands at %d, flags at %d positions" % (k, d_tt.sum(), d_code.sum(), d_fl.sum()))
                    for nm, dm in (("operands", d_code), ("flags", d_fl), ("tables", d_tt)):
                        for j in np.nonzero(dm)[0][:4].tolist():
                            print("    %s @%d: node %d eval=%s tt=%x code=%s fl=%d | ref node %d tt=%x code=%s fl=%d" % (
                                nm, j, g[j], g[j] < n_eval, TTn[g[j]], code[j].tolist(), fl[g[j]], ref_c[3][j], ref_c[0][j], ref_c[1][j].tolist(), ref_c[2][j]))
        best = (0, 0)
        k = 0
        while k < len(rng):
            j = k
            while j + 1 < len(rng) and digests[j + 1] == digests[k]:
                j += 1
            if j - k + 1 > best[1]:
                best = (k, j - k + 1)
            k = j + 1
        k0, K = best
        if flags["audit"]:
            # the audit walks the table the main program's loop wrote: the same iterations or none
            k0, K = main_loop["first"] - 1, main_loop["K"]
            if len(set(digests[k0:k0 + K])) != 1:
                continue
        L = bounds[k0 + 2] - bounds[k0 + 1]
        if dbg:
            print("loop candidate: %d instances, gates per instance %s, digests %s -> run of %d from %d" %
                  (len(rng), [bounds[k + 2] - bounds[k + 1] for k in range(len(rng))], [d.hex()[:6] for d in digests], K, k0))
        if K < 2 or (L < LOOP_MIN_BODY_GATES and not flags["audit"]) or L == 0:
            continue
        ext = np.stack([exts[k0 + i] for i in range(K)]) if len(exts[k0]) else np.zeros((K, 0), dtype=np.int64)
        return order, seg, bounds, {"first": k0 + 1, "K": K, "ext": ext}
    raise AssertionError("unreachable")


STORE_PACE = 0                  # 0: a row store where the value is produced (measured best); n: at most one per n instructions - lower_jit
STORE_PENDING_MAX = 48          # values whose row store is still to be issued


PREFETCH = 192                  # gates of look-ahead for values that come back from memory (round 6: 384 -> 192: 13.1 -> 12.2 ms and
                                # 13.3 -> 13.0 ms on two boxes, profiles/r06c / r06d_jit_variants.txt; 768: 14.0; 128: worse again)


def lower_jit(net, fc, n_vgpr: int = 256, n_agpr: int = 256, prefetch: int = PREFETCH, fuse_check: bool = True, hoist: int = 256, audit_of=None,
              loop: bool = True, store_pace=None, store_pending=None):
    """BitNet (bitblast.py) -> JitProgram with IR.  Returns None when there is nothing to evaluate.
    audit_of = the program lowered from the same network: lower the STAND-ALONE AUDIT of its table instead - the gates of the
    R1CS check alone, their wires LOADED from the rows that program stored (one coalesced 256-byte row per wire and wave: the
    table's own layout), nothing evaluated, nothing stored but scratch behind the table.  `cw_check_r1cs` runs it when the
    caller asks for an audit (CW_R1CS_AUDIT=1) or may have changed the table (cw_device_bits): the general check kernels
    read 8 bytes out of every 256-byte row in this layout (271 ms for 2^21 instances of Sha256(2048)).
    store_pace: PACED row stores, an experiment that is kept switched off (0).  A signal's row is stored right behind the gate
    that produces it: SHA-256's adders then issue 20-30 stores within a hundred instructions and none for the next two hundred
    (a third of all 128-instruction windows have no store, the busiest tenth 27+), and timing-only variants
    (profiles/r06c_jit_variants.txt) show the stores costing the kernel exactly their own stream time (13.1 ms with, 6.3 ms
    without, the store stream alone 6.35) although a uniform stream of the same density overlaps with VALU work completely
    (tools/ubench_loop.hip).  With store_pace = n a produced value waits in its register (VGPR or AccVGPR:
    `buffer_store_dword` takes either) and ONE store is issued every n instructions (a register that is wanted back, more than
    STORE_PENDING_MAX waiting values or a loop boundary flush earlier).  MEASURED (profiles/r06d_jit_variants.txt): pace 8 / 10 /
    12 = 13.38 / 13.44 / 13.50 ms against 13.28 unpaced - smoothing the stream does not help (bursts of consecutive rows are what
    the memory system likes), so burstiness is not why the stores do not overlap.
    loop: emit ONE body for the instances of a repeated template and run it once per instance (`_plan_segments`; needs a
    network built with bitblast(ports=, marks=) from `instance_ports(fc)`): code size follows the templates, not the circuit -
    the reference's own structure (template.rs:160-474, loop_bucket.rs:77).  Inside the body a row is addressed relative to
    the iteration's base, a value from outside the iteration through a per-iteration table of row offsets (IR "ldx")."""
    n_eval = len(net.tt)
    audit = audit_of is not None
    if audit and not (fuse_check and fc.constraints and audit_of.check_complete):
        return None
    if fuse_check and fc.constraints:
        # (the audit program of the same network - lower_jit(audit_of=) right after the main one - reuses the gates of the check:
        # building them is a quarter of the lowering of the 1M-constraint SHA-256)
        cached = getattr(net, "_check_cache", None)
        if cached is not None and cached[0] is fc:
            G, viol, cst = cached[1]
        else:
            G, viol, cst = build_check(net, fc)
            net._check_cache = (fc, (G, viol, cst))
    else:
        G, viol, cst = _Gates(n_eval), [], {"trivial": 0, "lut": 0, "int": 0, "unchecked": len(fc.constraints)}
    TT = list(net.tt) + G.tt
    A = list(net.a) + G.a
    B = list(net.b) + G.b
    C = list(net.c) + G.c
    n_nodes = len(TT)
    sn = np.asarray(net.sig_node, dtype=np.int64)
    is_signal = np.zeros(n_nodes, dtype=bool)
    is_signal[sn] = True
    is_gate = [t <= 0xFF for t in TT]
    is_gate[0] = is_gate[1] = False
    ports = _resolve_ports(net)                # port -> the node whose row it is read from
    al = list(range(n_nodes))
    for p_, x in ports.items():
        al[p_] = x
    if audit:
        live = [False] * n_nodes               # the check's own gates only; their wires are rows of the table
        for x in viol:
            live[x] = True
        for nid in range(n_nodes - 1, n_eval - 1, -1):
            if live[nid] and is_gate[nid]:
                live[A[nid]] = live[B[nid]] = live[C[nid]] = True
    else:
        live = is_signal.tolist()
        for x in net.asserts:
            live[x] = True
        for x in viol:
            live[x] = True
        for x in ports.values():
            live[x] = True                     # (what a port carries: the values wired to a component's inputs are signals anyway)
        for nid in range(n_nodes - 1, 1, -1):
            if live[nid] and is_gate[nid]:
                live[A[nid]] = live[B[nid]] = live[C[nid]] = True
    assert_set = set() if audit else set(int(x) for x in net.asserts if x > 1)
    viol_set = set(int(x) for x in viol if x > 1)
    const_assert = (not audit) and any(x == 1 for x in net.asserts)      # an assertion that is constant true-violation: every instance falls back
    const_viol = any(x == 1 for x in viol)
    if audit and any(x < n_eval and x > 1 for x in viol_set):
        return None                                # (a violation value that is an evaluation gate itself: not a shape build_check makes)
    fa = np.zeros(n_nodes, dtype=bool)
    fa[list(assert_set)] = True
    fv = np.zeros(n_nodes, dtype=bool)
    fv[list(viol_set)] = True
    planned = _plan_segments(net, fc, n_eval, TT, A, B, C, is_gate, live, G,
                             {"audit": audit, "signal": is_signal, "assert": fa, "viol": fv}, hoist,
                             loop and (not audit or getattr(audit_of, "loop", None) is not None), al,
                             getattr(audit_of, "loop", None) if audit else None,
                             np.asarray(audit_of.node_slot, dtype=np.int64) if audit else None)
    if planned is None:
        return None
    order_all, seg_of, bounds, lp = planned
    # the plan the allocator walks: prologue (+ instances in front of the loop) | ONE instance of the loop | the rest
    must_store = is_signal.copy()
    gm = np.asarray(is_gate, dtype=bool)
    al_n = np.asarray(al, dtype=np.int64)
    if ports and not audit:
        srcs = np.fromiter(ports.values(), dtype=np.int64, count=len(ports))
        must_store[srcs[gm[srcs]]] = True          # a port is read from its source's row
    if lp is not None:
        s0, K = lp["first"], lp["K"]
        b0, b1, b2 = bounds[s0], bounds[s0 + 1], bounds[s0 + K]
        L_body = b1 - b0
        order = np.concatenate([order_all[:b1], order_all[b2:]]).tolist()
        bodies = order_all[b0:b2].reshape(K, L_body)
        # a value of the loop that something outside its own iteration reads must have a row - in EVERY iteration alike
        An, Bn, Cn = (np.asarray(x, dtype=np.int64) for x in (A, B, C))
        users = order_all
        used_out = np.zeros(n_nodes, dtype=bool)
        for Xn in (An, Bn, Cn):
            o = al_n[Xn[users]]                                 # (through ports: the gate whose row is read)
            m = gm[o] & (seg_of[o] >= s0) & (seg_of[o] < s0 + K) & (seg_of[o] != seg_of[users])
            used_out[o[m]] = True
        force = used_out[bodies].any(axis=0)
        must_store[bodies[0][force]] = True
        ext_nodes = lp["ext"]                                   # [K][n_ext]
        ext_src = al_n[ext_nodes]
        must_store[ext_src[gm[ext_src]]] = True                 # prologue values the loop reads: through their rows
        ext_index = {int(nd): e for e, nd in enumerate(ext_nodes[0].tolist())}
        body_seg = s0
    else:
        order = order_all.tolist()
        b0 = b1 = -1
        K = 1
        ext_index = {}
        body_seg = -1
    n_ops = len(order)
    seg_l = seg_of.tolist()
    must_l = must_store.tolist()

    # uses per node (positions in `order`), consumed front to back
    uses = [None] * n_nodes
    for p, g in enumerate(order):
        for o in (A[g], B[g], C[g]):
            if o > 1:
                u = uses[o]
                if u is None:
                    uses[o] = [p]
                elif u[-1] != p:
                    u.append(p)
    ptr = [0] * n_nodes

    def next_use(n):
        u = uses[n]
        if u is None:
            return INF
        i = ptr[n]
        return u[i] if i < len(u) else INF

    # ---- state ---------------------------------------------------------------------------------------------------------
    V_FIRST = 3                                   # v0 = lane * 4, v1 = fallback accumulator, v2 = R1CS accumulator
    free_v = list(range(n_vgpr - 1, V_FIRST - 1, -1))
    free_a = list(range(n_agpr - 1, -1, -1))
    loc_v = [-1] * n_nodes
    loc_a = [-1] * n_nodes
    mem_slot = [-1] * n_nodes                     # row holding the value (inputs, stored signals, scratch)
    st_idx = [-1] * n_nodes                       # in-order index of the store that wrote mem_slot (-1: written before the kernel)
    pend = [-1] * n_nodes                         # in-order index of a load in flight into loc_v
    vheap, aheap = [], []                         # (-next use, node)
    if audit:
        ns_ = np.asarray(audit_of.node_slot, dtype=np.int64)
        assert len(ns_) == n_eval
        for nid in np.nonzero(ns_ >= 0)[0].tolist():
            mem_slot[nid] = int(ns_[nid])
        next_slot = int(audit_of.n_slots)          # scratch rows of the audit live behind the table's own
    else:
        for s, nid in net.input_node.items():
            mem_slot[nid] = IN_BASE + (s - fc.main_input_start)
        mem_slot[0], mem_slot[1] = 0, 1            # the constant rows (a port may carry a constant)
        next_slot = IN_BASE + fc.n_main_inputs
    ir = []
    emit = ir.append
    if store_pace is None:
        store_pace = int(os.environ.get("CW_JIT_STORE_PACE", STORE_PACE))
    pending_max = int(os.environ.get("CW_JIT_STORE_PENDING", STORE_PENDING_MAX)) if store_pending is None else store_pending
    from collections import deque
    pending = deque()                             # produced values whose row store has not been issued yet (paced stores)
    is_pending = [False] * n_nodes
    n_pending = 0
    last_store_at = -(1 << 30)                    # IR position of the latest row store
    vm_issued = 0
    vm_done = -1                                  # every memory operation with index <= vm_done has completed
    stats = {"gates": 0, "stores": 0, "prefetched": 0, "late_loads": 0, "agpr_writes": 0, "agpr_reads": 0, "scratch_stores": 0,
             "waits": 0, "loads_for_check": 0}
    dyn = dict(stats)                             # the same counts as EXECUTED (the loop's body counts once per iteration)
    in_body = False
    loop_base = 0

    def count(k_, n=1):
        stats[k_] += n
        dyn[k_] += n * (K if in_body else 1)

    def wait_for(idx):
        nonlocal vm_done
        if idx <= vm_done:
            return
        n = vm_issued - 1 - idx
        if n > 63:
            n = 63
        emit(("w", n))
        count("waits")
        vm_done = vm_issued - 1 - n

    def issue_load(node, v):
        nonlocal vm_issued
        src = al[node]                            # (a port: the row of the value it carries)
        si = st_idx[src]
        if si >= 0:
            wait_for(si)                          # the row was written by this wave: the store must have completed
        if not in_body:
            emit(("ld", v, mem_slot[src]))
        elif seg_l[node] == body_seg and is_gate[node]:
            # a row of this iteration (the audit: a wire the main program's iteration stored, or the audit's own scratch)
            emit(("ldL", v, mem_slot[node] - loop_base))
        else:
            emit(("ldx", v, ext_index[node]))     # a value from outside the iteration: its row comes from the iteration's table
        pend[node] = vm_issued
        vm_issued += 1

    def issue_store(kind, reg, node):
        nonlocal vm_issued, next_slot, last_store_at, n_pending
        if is_pending[node]:
            is_pending[node] = False              # (its paced store is this one)
            n_pending -= 1
        last_store_at = len(ir)
        mem_slot[node] = next_slot
        next_slot += 1
        if in_body:
            emit((kind + "L", reg, mem_slot[node] - loop_base))
        else:
            emit((kind, reg, mem_slot[node]))
        st_idx[node] = vm_issued
        vm_issued += 1

    def alloc_a(nu_in):
        """an AccVGPR for a value whose next use is nu_in, or -1 when that value should go to memory instead"""
        if free_a:
            return free_a.pop()
        while aheap:
            nnu, nd = aheap[0]
            if loc_a[nd] < 0 or next_use(nd) != -nnu:
                heapq.heappop(aheap)
                continue
            break
        if not aheap:
            return -1
        nnu, nd = aheap[0]
        if -nnu <= nu_in:
            return -1                             # everything in the AccVGPRs is needed sooner
        heapq.heappop(aheap)
        a = loc_a[nd]
        if mem_slot[nd] < 0:
            was_pending = is_pending[nd]
            issue_store("sta", a, nd)
            count("stores" if (was_pending and is_signal[nd]) else "scratch_stores")
        loc_a[nd] = -1
        return a

    def evict_v(pin):
        stash = []
        victim = -1
        while vheap:
            nnu, nd = heapq.heappop(vheap)
            if loc_v[nd] < 0 or next_use(nd) != -nnu:
                continue
            if nd in pin or pend[nd] >= 0:
                stash.append((nnu, nd))
                continue
            victim = nd
            break
        if victim < 0:                            # only pinned / in-flight values left: take an in-flight one
            for k, (nnu, nd) in enumerate(stash):
                if nd not in pin:
                    victim = nd
                    stash.pop(k)
                    wait_for(pend[nd])
                    pend[nd] = -1
                    break
        for e in stash:
            heapq.heappush(vheap, e)
        if victim < 0:
            raise RuntimeError("register allocation: no evictable VGPR")
        if is_pending[victim]:
            flush_pending(victim)                 # the store it was waiting for, now; a dead value is gone with it
            if loc_v[victim] < 0:
                return free_v.pop()
        v = loc_v[victim]
        loc_v[victim] = -1
        if loc_a[victim] < 0:
            # second level: the AccVGPRs hold the evicted values that are needed soonest (a value that also has a row in the
            # bit table may simply be dropped, but a reload costs a memory instruction, its wait and a trip to the L2)
            nu = next_use(victim)
            a = alloc_a(nu)
            if a >= 0:
                emit(("aw", a, v))
                count("agpr_writes")
                loc_a[victim] = a
                heapq.heappush(aheap, (-nu, victim))
            elif mem_slot[victim] < 0:
                issue_store("st", v, victim)
                count("scratch_stores")
        return v

    def alloc_v(pin):
        if free_v:
            return free_v.pop()
        return evict_v(pin)

    def release(node):
        if is_pending[node]:
            # its row store is still to come: the register stays, first in line when one is wanted back
            if loc_v[node] >= 0:
                heapq.heappush(vheap, (-INF, node))
            else:
                heapq.heappush(aheap, (-INF, node))
            return
        v = loc_v[node]
        if v >= 0:
            if pend[node] >= 0:                   # (cannot happen for a used value: kept for safety)
                wait_for(pend[node])
                pend[node] = -1
            free_v.append(v)
            loc_v[node] = -1
        a = loc_a[node]
        if a >= 0:
            free_a.append(a)
            loc_a[node] = -1

    def flush_pending(node):
        """issue the row store of a waiting value now (from the VGPR or the AccVGPR it lives in); a value nothing reads again
        gives its register back"""
        v, a = loc_v[node], loc_a[node]
        if v >= 0:
            if pend[node] >= 0:                   # (cannot be: a waiting value was computed, not loaded)
                wait_for(pend[node])
                pend[node] = -1
            issue_store("st", v, node)
        else:
            assert a >= 0, "a value waiting for its store lives in no register"
            issue_store("sta", a, node)
        count("stores" if is_signal[node] else "scratch_stores")
        if next_use(node) == INF:
            if v >= 0:
                free_v.append(v)
                loc_v[node] = -1
            if a >= 0:
                free_a.append(a)
                loc_a[node] = -1

    def pace_stores(everything=False):
        """one waiting store if the latest is `store_pace` instructions back; more while too many wait; all at a boundary"""
        while pending:
            nd = pending[0]
            if not is_pending[nd]:
                pending.popleft()                 # (flushed early: its register was wanted)
                continue
            if not (everything or n_pending > pending_max or len(ir) - last_store_at >= store_pace):
                break
            pending.popleft()
            flush_pending(nd)

    empty = frozenset()
    held_viol = [-1]                              # VGPR of a violation value waiting for its partner (not in any heap: never evicted)

    def boundary():
        """loop entry / exit: nothing lives in a register across it.  Values that are read again get a row (if they have none),
        every memory operation completes, the register files start empty."""
        nonlocal free_v, free_a, vm_done
        pace_stores(everything=True)
        if held_viol[0] >= 0:
            emit(("acc", 2, held_viol[0]))
            held_viol[0] = -1
        for heap, loc, kind in ((vheap, loc_v, "st"), (aheap, loc_a, "sta")):
            for nnu, nd in heap:
                if loc[nd] >= 0 and next_use(nd) == -nnu and mem_slot[nd] < 0:
                    if kind == "st" and pend[nd] >= 0:
                        wait_for(pend[nd])
                        pend[nd] = -1
                    issue_store(kind, loc[nd], nd)
                    count("scratch_stores")
        for heap, loc in ((vheap, loc_v), (aheap, loc_a)):
            for _, nd in heap:
                loc[nd] = -1
                pend[nd] = -1
            del heap[:]
        if vm_issued:
            wait_for(vm_issued - 1)
        free_v = list(range(n_vgpr - 1, V_FIRST - 1, -1))
        free_a = list(range(n_agpr - 1, -1, -1))

    loop_at = -1                                  # position of the ("loop", ...) instruction in the IR

    loop_info = [None]
    outer_next = [0]

    def end_loop():
        nonlocal in_body, next_slot
        boundary()
        in_body = False
        if audit:
            # the audit's scratch rows sit in the spare rows the main program left at the end of every iteration
            R = audit_of.loop["R"]
            if next_slot > loop_base + R:
                raise _AuditTooBig()
            next_slot = outer_next[0]
        else:
            used = next_slot - loop_base
            R = used + (AUDIT_SCRATCH_ROWS if (fuse_check and fc.constraints) else 0)
            # the rows of the other iterations: the same local row, R rows further per iteration
            ms0 = np.asarray([mem_slot[int(x)] for x in bodies[0]], dtype=np.int64)
            stored_ = ms0 >= 0
            for i in range(1, K):
                for nd, r in zip(bodies[i][stored_].tolist(), (ms0[stored_] + i * R).tolist()):
                    mem_slot[nd] = r
            next_slot = loop_base + K * R
            loop_info[0] = {"base": loop_base, "R": R, "used": used, "first": lp["first"], "K": K}
        tab = np.asarray([[mem_slot[al[int(nd)]] for nd in row] for row in ext_nodes], dtype=np.int64).reshape(K, -1)
        assert (tab >= 0).all(), "a value the loop reads from outside its iteration has no row"
        ir[loop_at] = ("loop", K, R, loop_base, tab.astype(np.uint32))
        emit(("endloop",))

    for p in range(n_ops):
        if p == b0:
            boundary()
            if audit:
                outer_next[0] = next_slot
                loop_base = audit_of.loop["base"]
                next_slot = loop_base + audit_of.loop["used"]
            else:
                loop_base = next_slot
            loop_at = len(ir)
            emit(("loop",))                       # (filled in at the end of the body)
            in_body = True
        elif p == b1:
            end_loop()
        if pending:
            pace_stores()
        region = 0 if p < b0 else 1 if p < b1 else 2
        # -- prefetch what the gate PREFETCH positions ahead reads from memory; a second look a quarter of that distance ahead
        # catches values that were resident at the first look and have been dropped since (they have a row: dropping is free)
        for pf in (p + prefetch, p + prefetch // 4):
            if pf >= n_ops or pf == p or (0 if pf < b0 else 1 if pf < b1 else 2) != region:
                continue
            g2 = order[pf]
            for o in (A[g2], B[g2], C[g2]):
                if o > 1 and loc_v[o] < 0 and loc_a[o] < 0 and mem_slot[al[o]] >= 0:
                    v = alloc_v(empty)
                    loc_v[o] = v
                    issue_load(o, v)
                    heapq.heappush(vheap, (-next_use(o), o))
                    count("prefetched")
                    if g2 >= n_eval:
                        count("loads_for_check")
        g = order[p]
        ops = (A[g], B[g], C[g])
        pin = set(o for o in ops if o > 1)
        for o in pin:
            if loc_v[o] < 0:
                if al[o] != o and is_pending[al[o]]:
                    flush_pending(al[o])           # a port reads its source's ROW: the row store it is waiting for, now
                v = alloc_v(pin)
                loc_v[o] = v
                if loc_a[o] >= 0:
                    emit(("ar", v, loc_a[o]))
                    count("agpr_reads")
                    free_a.append(loc_a[o])
                    loc_a[o] = -1
                elif mem_slot[al[o]] >= 0:
                    issue_load(o, v)
                    count("late_loads")
                    if g >= n_eval:
                        count("loads_for_check")
                elif audit:
                    return None                    # a wire of the check that the evaluation never stored: no audit program
                else:
                    raise RuntimeError("value %d is nowhere" % o)
        for o in pin:
            if pend[o] >= 0:
                wait_for(pend[o])
                pend[o] = -1
        srcs = []
        for o in (ops[2], ops[1], ops[0]):            # v_bitop3 sources (s0, s1, s2) = (c, b, a)
            srcs.append(-1 if o == 0 else -2 if o == 1 else loc_v[o])
        for o in pin:
            u = uses[o]
            i = ptr[o]
            while i < len(u) and u[i] <= p:
                i += 1
            ptr[o] = i
            if i >= len(u):
                release(o)
            else:
                heapq.heappush(vheap, (-u[i], o))
        stored = must_l[g]
        nu = next_use(g)
        d = alloc_v(pin)
        emit(("g", d, srcs[0], srcs[1], srcs[2], TT[g]))
        count("gates")
        loc_v[g] = d
        if stored:
            if store_pace > 0:
                pending.append(g)                 # its row store follows when the store stream has room (pace_stores)
                is_pending[g] = True
                n_pending += 1
            else:
                issue_store("st", d, g)
                count("stores" if is_signal[g] else "scratch_stores")
        if g in assert_set:
            emit(("acc", 1, d))
        if g in viol_set:
            # two violation values per v_or3_b32: the first of a pair waits in its register (pinned by `held`)
            if held_viol[0] < 0 and nu == INF:
                held_viol[0] = d
                continue
            if held_viol[0] >= 0:
                emit(("acc3", 2, held_viol[0], d))
                free_v.append(held_viol[0])
                held_viol[0] = -1
            else:
                emit(("acc", 2, d))
        if nu == INF:
            if is_pending[g]:
                heapq.heappush(vheap, (-INF, g))  # nothing reads it again, but its row store is still to come
            else:
                free_v.append(d)
                loc_v[g] = -1
        else:
            heapq.heappush(vheap, (-nu, g))
    pace_stores(everything=True)
    if lp is not None and n_ops == b1:                # (the loop ends the program: no gate behind it ran the exit code above)
        end_loop()
    if held_viol[0] >= 0:
        emit(("acc", 2, held_viol[0]))
    if const_assert:
        emit(("accc", 1))
    if const_viol:
        emit(("accc", 2))

    jp = JitProgram()
    jp.ir = ir
    jp.n_slots = (next_slot + 15) // 16 * 16
    jp.loop = loop_info[0]
    if audit:
        jp.sig_slot = audit_of.sig_slot
        jp.n_signals, jp.n_inputs, jp.input_start = audit_of.n_signals, audit_of.n_inputs, audit_of.input_start
        jp.n_vgpr, jp.n_agpr = n_vgpr, n_agpr
        jp.check_complete = True
        stats["instructions"] = len(ir)
        stats["slots"] = jp.n_slots
        jp.stats = stats
        jp.is_audit = True
        return jp
    for p_, x in ports.items():
        mem_slot[p_] = mem_slot[x]                # the signal of a port lives in the row of the value it carries
    sig_slot = np.zeros(fc.n_signals, dtype=np.uint32)
    ms = np.asarray(mem_slot, dtype=np.int64)
    jp.node_slot = ms[:n_eval].copy()             # evaluation node -> row (-1: never stored), for the audit program (audit_of=)
    node_slot = ms[sn]
    node_slot[sn == 0] = 0
    node_slot[sn == 1] = 1
    assert (node_slot >= 0).all(), "a signal value has no row"
    sig_slot[:] = node_slot
    jp.sig_slot = sig_slot
    jp.n_signals = fc.n_signals
    jp.n_inputs = fc.n_main_inputs
    jp.input_start = fc.main_input_start
    jp.n_vgpr, jp.n_agpr = n_vgpr, n_agpr
    jp.check_complete = bool(fuse_check and cst["unchecked"] == 0)
    stats.update({"check_" + k: v for k, v in cst.items()})
    stats["check_gates"] = len(G.tt)
    stats["instructions"] = len(ir)
    stats["slots"] = jp.n_slots
    stats["nodes"] = n_nodes
    if lp is not None:
        body_len = ir.index(("endloop",)) - loop_at - 1
        stats["loop"] = {"iterations": K, "body_instructions": body_len, "rows_per_iteration": int(ir[loop_at][2]),
                         "external_values": int(ir[loop_at][4].shape[1]), "segments": len(bounds) - 1}
        dyn["instructions"] = len(ir) + (K - 1) * body_len
        stats["executed"] = dyn                   # what a wave EXECUTES (bench.py prices the kernel with these)
    jp.stats = stats
    return jp


# ---- assembly text ---------------------------------------------------------------------------------------------------------
_VOP2 = {0b1000: "v_and_b32", 0b0110: "v_xor_b32", 0b1110: "v_or_b32", 0b1001: "v_xnor_b32"}


def gate_asm(ins) -> str:
    """one gate ('g', dst, s0, s1, s2, table; table bit (s0 << 2 | s1 << 1 | s2); operand -1 = constant 0, -2 = constant ones).
    The kernel is straight-line code that is FETCHED rather than cached (12.9 MB per wave for the 1 M-constraint SHA-256), so
    bytes of code are time: a gate with one constant operand that is AND / OR / XOR / XNOR of the other two (40 % of SHA-256's
    gates: the partial terms of Ch / Maj / the adders' carries against a known bit) takes the 4-byte VOP2 form instead of the
    8-byte v_bitop3_b32 (-14 % code).  `tests/test_bitjit.py` evaluates the printed text against the table."""
    _, dst, a, b, c, tt = ins
    ops = (a, b, c)
    var = [x for x in ops if x >= 0]
    if len(var) == 2 and var[0] != var[1]:
        red = 0
        for m in range(4):
            bits = iter(((m >> 1) & 1, m & 1))
            full = 0
            for pos, x in enumerate(ops):
                v = 0 if x == -1 else 1 if x == -2 else next(bits)
                full |= v << (2 - pos)
            red |= ((tt >> full) & 1) << m
        op = _VOP2.get(red)
        if op is not None:
            return "  %s v%d, v%d, v%d\n" % (op, dst, var[0], var[1])

    def opnd(x):
        return "0" if x == -1 else "-1" if x == -2 else "v%d" % x
    return "  v_bitop3_b32 v%d, %s, %s, %s bitop3:0x%x\n" % (dst, opnd(a), opnd(b), opnd(c), tt)


S_ITER, S_COUNT, S_TOP, S_TAB = 40, 41, 42, 44       # loop state: iteration base (bytes), iterations left, loop top pc, table cursor
S_EXT0, EXT_SLOTS, EXT_BATCH = 48, 6, 8                # row offsets of external values: 6 x s_load_dwordx8 (s48 .. s95)
N_SGPR = 96


def to_asm(jp: JitProgram) -> str:
    """gfx950 assembly of the program.  Kernel arguments: bit table, fallback masks, R1CS flags (three pointers); one
    wave (workgroup of 64) per chunk of 2 048 instances.
    A loop (IR "loop" ... "endloop") runs its body once per instance of the repeated template: rows of the body are addressed
    relative to the iteration's base (s40, advanced by the body's rows), values from outside the iteration through the
    iteration's table of row offsets, which sits behind the code (s[44:45] walks it; entries come in through s_load_dwordx8
    into six rotating octets of SGPRs) - the reference's `signalValues[mySignalStart + ...]` (template.rs:297-304)."""
    chunk_bytes = jp.n_slots * ROW_BYTES
    assert chunk_bytes < (1 << 32)
    L = []
    add = L.append
    add('.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n.text\n.globl %s\n.p2align 8\n.type %s,@function\n%s:\n'
        % (KERNEL_NAME, KERNEL_NAME, KERNEL_NAME))
    # s[0:1] kernarg, s2 workgroup id; s[4:5] T, s[6:7] fallback masks, s[8:9] R1CS flags; s[12:15] buffer descriptor of the
    # chunk; s16 = page of the store stream; s[20 .. 20 + N_PAGE_SGPRS) = pages of loads
    add("  s_load_dwordx4 s[4:7], s[0:1], 0x0\n  s_load_dwordx2 s[8:9], s[0:1], 0x10\n"
        "  v_lshlrev_b32 v0, 2, v0\n  v_mov_b32 v1, 0\n  v_mov_b32 v2, 0\n"
        "  s_mov_b32 s10, 0x%x\n  s_waitcnt lgkmcnt(0)\n"
        "  s_mul_i32 s11, s2, s10\n  s_mul_hi_u32 s17, s2, s10\n"
        "  s_add_u32 s12, s4, s11\n  s_addc_u32 s13, s5, s17\n  s_and_b32 s13, s13, 0xffff\n"
        "  s_mov_b32 s14, s10\n  s_mov_b32 s15, 0x00020000\n" % chunk_bytes)
    st_page = [-1]
    pages = {}                   # page -> sgpr
    lru = []                     # pages, most recent last
    free_s = list(range(20 + N_PAGE_SGPRS - 1, 19, -1))
    in_loop = [False]
    ext_slot = {}                # batch of the iteration's table -> first sgpr
    ext_lru = []
    tables = []                  # the loops' tables, printed behind the code
    loop_no = [0]

    def reset_pages():
        st_page[0] = -1
        pages.clear()
        del lru[:]
        del free_s[:]
        free_s.extend(range(20 + N_PAGE_SGPRS - 1, 19, -1))
        ext_slot.clear()
        del ext_lru[:]

    def page_set(s_, pg):
        if in_loop[0]:
            add("  s_add_u32 s%d, s%d, 0x%x\n" % (s_, S_ITER, pg * PAGE))
        else:
            add("  s_mov_b32 s%d, 0x%x\n" % (s_, pg * PAGE))

    def load_page(pg):
        s_ = pages.get(pg)
        if s_ is not None:
            if lru[-1] != pg:
                lru.remove(pg)
                lru.append(pg)
            return s_
        if free_s:
            s_ = free_s.pop()
        else:
            old = lru.pop(0)
            s_ = pages.pop(old)
        pages[pg] = s_
        lru.append(pg)
        page_set(s_, pg)
        return s_

    def ext_sgpr(e):
        b = e // EXT_BATCH
        s_ = ext_slot.get(b)
        if s_ is None:
            if len(ext_slot) < EXT_SLOTS:
                s_ = S_EXT0 + EXT_BATCH * len(ext_slot)
            else:
                old = ext_lru.pop(0)
                s_ = ext_slot.pop(old)
            ext_slot[b] = s_
            add("  s_load_dwordx8 s[%d:%d], s[%d:%d], 0x%x\n  s_waitcnt lgkmcnt(0)\n" % (s_, s_ + EXT_BATCH - 1, S_TAB, S_TAB + 1, b * EXT_BATCH * 4))
        elif ext_lru[-1] != b:
            ext_lru.remove(b)
        if not ext_lru or ext_lru[-1] != b:
            ext_lru.append(b)
        return s_ + e % EXT_BATCH

    for ins in jp.ir:
        k = ins[0]
        if k == "g":
            add(gate_asm(ins))
        elif k in ("st", "sta", "stL", "staL"):
            off = ins[2] * ROW_BYTES
            pg = off // PAGE
            if pg != st_page[0]:
                page_set(16, pg)
                st_page[0] = pg
            add("  buffer_store_dword %s%d, v0, s[12:15], s16 offen offset:%d nt\n" % ("v" if k in ("st", "stL") else "a", ins[1], off % PAGE))
        elif k == "ld" or k == "ldL":
            off = ins[2] * ROW_BYTES
            s_ = load_page(off // PAGE)
            add("  buffer_load_dword v%d, v0, s[12:15], s%d offen offset:%d\n" % (ins[1], s_, off % PAGE))
        elif k == "ldx":
            add("  buffer_load_dword v%d, v0, s[12:15], s%d offen\n" % (ins[1], ext_sgpr(ins[2])))
        elif k == "w":
            add("  s_waitcnt vmcnt(%d)\n" % ins[1])
        elif k == "aw":
            add("  v_accvgpr_write_b32 a%d, v%d\n" % (ins[1], ins[2]))
        elif k == "ar":
            add("  v_accvgpr_read_b32 v%d, a%d\n" % (ins[1], ins[2]))
        elif k == "acc":
            add("  v_or_b32 v%d, v%d, v%d\n" % (ins[1], ins[1], ins[2]))
        elif k == "acc3":
            add("  v_or3_b32 v%d, v%d, v%d, v%d\n" % (ins[1], ins[1], ins[2], ins[3]))
        elif k == "accc":
            add("  v_mov_b32 v%d, -1\n" % ins[1])
        elif k == "loop":
            _, K, R, base, tab = ins
            n_ext_pad = max(EXT_BATCH, (tab.shape[1] + EXT_BATCH - 1) // EXT_BATCH * EXT_BATCH)
            tables.append((loop_no[0], n_ext_pad, tab))
            # s[44:45] = address of the table (pc-relative: it sits behind s_endpgm in this section); s[42:43] = the loop's top
            add("  s_mov_b32 s%d, 0x%x\n  s_mov_b32 s%d, %d\n  s_getpc_b64 s[%d:%d]\n.Lanchor%d:\n"
                "  s_add_u32 s%d, s%d, cw_ext_tab%d-.Lanchor%d\n  s_addc_u32 s%d, s%d, 0\n  s_getpc_b64 s[%d:%d]\n"
                % (S_ITER, base * ROW_BYTES, S_COUNT, K, S_TAB, S_TAB + 1, loop_no[0], S_TAB, S_TAB, loop_no[0], loop_no[0], S_TAB + 1, S_TAB + 1,
                   S_TOP, S_TOP + 1))
            in_loop[0] = True
            in_loop.append((R, n_ext_pad))
            reset_pages()
        elif k == "endloop":
            R, n_ext_pad = in_loop.pop()
            add("  s_add_u32 s%d, s%d, 0x%x\n  s_add_u32 s%d, s%d, 0x%x\n  s_addc_u32 s%d, s%d, 0\n"
                "  s_sub_u32 s%d, s%d, 1\n  s_cmp_lg_u32 s%d, 0\n  s_cbranch_scc0 .Ldone%d\n  s_setpc_b64 s[%d:%d]\n.Ldone%d:\n"
                % (S_ITER, S_ITER, R * ROW_BYTES, S_TAB, S_TAB, n_ext_pad * 4, S_TAB + 1, S_TAB + 1,
                   S_COUNT, S_COUNT, S_COUNT, loop_no[0], S_TOP, S_TOP + 1, loop_no[0]))
            in_loop[0] = False
            loop_no[0] += 1
            reset_pages()
        else:
            raise ValueError(k)
    # flags: one dword per lane = 32 instances; fallback masks / R1CS flags are uint64 per group of 64 instances, i.e. dword
    # chunk * 64 + lane of the array
    add("  s_lshl_b32 s17, s2, 8\n  v_add_u32 v0, s17, v0\n"
        "  global_atomic_or v0, v1, s[6:7]\n  global_atomic_or v0, v2, s[8:9]\n"
        "  s_endpgm\n")
    for no, n_ext_pad, tab in tables:
        add(".p2align 6\ncw_ext_tab%d:\n" % no)
        for row in tab:
            vals = [int(x) * ROW_BYTES for x in row] + [0] * (n_ext_pad - len(row))
            for i in range(0, len(vals), 16):
                add("  .long " + ", ".join("0x%x" % v for v in vals[i:i + 16]) + "\n")
    add(".Lend:\n.size %s, .Lend-%s\n" % (KERNEL_NAME, KERNEL_NAME))
    add(".rodata\n.p2align 6\n.amdhsa_kernel %s\n"
        "  .amdhsa_user_sgpr_kernarg_segment_ptr 1\n  .amdhsa_system_sgpr_workgroup_id_x 1\n  .amdhsa_system_vgpr_workitem_id 0\n"
        "  .amdhsa_next_free_vgpr %d\n  .amdhsa_accum_offset %d\n  .amdhsa_next_free_sgpr %d\n"
        "  .amdhsa_group_segment_fixed_size 0\n  .amdhsa_private_segment_fixed_size 0\n  .amdhsa_kernarg_size 24\n"
        ".end_amdhsa_kernel\n" % (KERNEL_NAME, jp.n_vgpr + jp.n_agpr, jp.n_vgpr, N_SGPR))
    add(".amdgpu_metadata\n---\namdhsa.version: [1, 2]\namdhsa.kernels:\n  - .name: %s\n    .symbol: %s.kd\n"
        "    .kernarg_segment_size: 24\n    .group_segment_fixed_size: 0\n    .private_segment_fixed_size: 0\n"
        "    .kernarg_segment_align: 8\n    .wavefront_size: 64\n    .sgpr_count: %d\n    .vgpr_count: %d\n    .agpr_count: %d\n"
        "    .max_flat_workgroup_size: 64\n    .args:\n"
        "      - {.size: 8, .offset: 0, .value_kind: global_buffer, .address_space: global}\n"
        "      - {.size: 8, .offset: 8, .value_kind: global_buffer, .address_space: global}\n"
        "      - {.size: 8, .offset: 16, .value_kind: global_buffer, .address_space: global}\n"
        "...\n.end_amdgpu_metadata\n" % (KERNEL_NAME, KERNEL_NAME, N_SGPR, jp.n_vgpr + jp.n_agpr, jp.n_agpr))
    return "".join(L)


def assemble(asm: str) -> bytes:
    """assembly text -> code object (ELF for hipModuleLoadData) with the ROCm LLVM assembler and linker"""
    llvm = _llvm_bin()
    with tempfile.TemporaryDirectory(prefix="cw_jit_") as d:
        s, o, co = os.path.join(d, "k.s"), os.path.join(d, "k.o"), os.path.join(d, "k.co")
        with open(s, "w") as f:
            f.write(asm)
        subprocess.run([os.path.join(llvm, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o],
                       check=True, capture_output=True)
        subprocess.run([os.path.join(llvm, "ld.lld"), "-shared", o, "-o", co], check=True, capture_output=True)
        return open(co, "rb").read()
