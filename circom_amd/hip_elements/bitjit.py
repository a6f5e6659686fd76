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

    def lin(d):
        """{signal: coef} -> (c0, {node: coef}) in signed representatives"""
        c0 = 0
        t = {}
        for s, cf in d.items():
            nd = int(sn[s]) if s else 1
            if nd == 0:
                continue
            cf = signed(cf)
            if nd == 1:
                c0 += cf
            else:
                t[nd] = t.get(nd, 0) + cf
        return signed(c0), {k: signed(v) for k, v in t.items() if v % q}

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
        if len(wires) <= LUT_MAX_WIRES:
            n = len(wires)
            tt = 0
            for m in range(1 << n):
                av, bv, cv = a0, b0, c0
                for j, w in enumerate(wires):
                    if (m >> j) & 1:
                        av += at.get(w, 0); bv += bt.get(w, 0); cv += ct.get(w, 0)
                if (av * bv - cv) % q:
                    tt |= 1 << m
            if tt == 0:
                st["trivial"] += 1
                continue
            v = G.lut(wires, tt)
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


def lower_jit(net, fc, n_vgpr: int = 256, n_agpr: int = 256, prefetch: int = 384, fuse_check: bool = True, hoist: int = 256, audit_of=None):
    """BitNet (bitblast.py) -> JitProgram with IR.  Returns None when there is nothing to evaluate.
    audit_of = the program lowered from the same network: lower the STAND-ALONE AUDIT of its table instead - the gates of the
    R1CS check alone, their wires LOADED from the rows that program stored (one coalesced 256-byte row per wire and wave: the
    table's own layout), nothing evaluated, nothing stored but scratch behind the table.  `cw_check_r1cs` runs it when the
    caller asks for an audit (CW_R1CS_AUDIT=1) or may have changed the table (cw_device_bits): the general check kernels
    read 8 bytes out of every 256-byte row in this layout (271 ms for 2^21 instances of Sha256(2048))."""
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
        for nid in range(n_nodes - 1, 1, -1):
            if live[nid] and is_gate[nid]:
                live[A[nid]] = live[B[nid]] = live[C[nid]] = True
    # order: evaluation gates in creation order (= program order of the witness code) - except that a gate whose operands
    # were ALL created long before it moves up behind the youngest of them (circomlib's SHA-256 computes a block twice: the
    # `<--` hint function first, then the constrained components, whose values are the hint's gates again plus a few of
    # their own - the `mid` products, partial sums: created a whole block later than everything they read and are read with);
    # every check gate sits where the youngest wire of its constraint is produced
    key = np.zeros(n_nodes, dtype=np.int64)
    kl = key.tolist()
    for nid in range(2, n_eval):
        if is_gate[nid]:
            m = max(A[nid], B[nid], C[nid])
            kl[nid] = nid if nid - m <= hoist else max(kl[A[nid]], kl[B[nid]], kl[C[nid]])
        else:
            kl[nid] = 0                        # inputs are there from the start
    for nid in range(n_eval, n_nodes):
        i = nid - n_eval
        kl[nid] = max(kl[G.rowkey[i]], kl[A[nid]], kl[B[nid]], kl[C[nid]])
    key = np.asarray(kl, dtype=np.int64)
    gates = np.array([i for i in range(n_eval if audit else 2, n_nodes) if is_gate[i] and live[i]], dtype=np.int64)
    if len(gates) == 0:
        return None
    order = gates[np.lexsort((gates, key[gates]))].tolist()
    n_ops = len(order)
    assert_set = set() if audit else set(int(x) for x in net.asserts if x > 1)
    viol_set = set(int(x) for x in viol if x > 1)
    const_assert = (not audit) and any(x == 1 for x in net.asserts)      # an assertion that is constant true-violation: every instance falls back
    const_viol = any(x == 1 for x in viol)
    if audit and any(x < n_eval and x > 1 for x in viol_set):
        return None                                # (a violation value that is an evaluation gate itself: not a shape build_check makes)

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
        next_slot = IN_BASE + fc.n_main_inputs
    ir = []
    emit = ir.append
    vm_issued = 0
    vm_done = -1                                  # every memory operation with index <= vm_done has completed
    stats = {"gates": 0, "stores": 0, "prefetched": 0, "late_loads": 0, "agpr_writes": 0, "agpr_reads": 0, "scratch_stores": 0,
             "waits": 0, "loads_for_check": 0}

    def wait_for(idx):
        nonlocal vm_done
        if idx <= vm_done:
            return
        n = vm_issued - 1 - idx
        if n > 63:
            n = 63
        emit(("w", n))
        stats["waits"] += 1
        vm_done = vm_issued - 1 - n

    def issue_load(node, v):
        nonlocal vm_issued
        si = st_idx[node]
        if si >= 0:
            wait_for(si)                          # the row was written by this wave: the store must have completed
        emit(("ld", v, mem_slot[node]))
        pend[node] = vm_issued
        vm_issued += 1

    def issue_store(kind, reg, node):
        nonlocal vm_issued, next_slot
        mem_slot[node] = next_slot
        next_slot += 1
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
            issue_store("sta", a, nd)
            stats["scratch_stores"] += 1
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
        v = loc_v[victim]
        loc_v[victim] = -1
        if loc_a[victim] < 0:
            # second level: the AccVGPRs hold the evicted values that are needed soonest (a value that also has a row in the
            # bit table may simply be dropped, but a reload costs a memory instruction, its wait and a trip to the L2)
            nu = next_use(victim)
            a = alloc_a(nu)
            if a >= 0:
                emit(("aw", a, v))
                stats["agpr_writes"] += 1
                loc_a[victim] = a
                heapq.heappush(aheap, (-nu, victim))
            elif mem_slot[victim] < 0:
                issue_store("st", v, victim)
                stats["scratch_stores"] += 1
        return v

    def alloc_v(pin):
        if free_v:
            return free_v.pop()
        return evict_v(pin)

    def release(node):
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

    empty = frozenset()
    held_viol = [-1]                              # VGPR of a violation value waiting for its partner (not in any heap: never evicted)
    for p in range(n_ops):
        # -- prefetch what the gate PREFETCH positions ahead reads from memory; a second look a quarter of that distance ahead
        # catches values that were resident at the first look and have been dropped since (they have a row: dropping is free)
        for pf in (p + prefetch, p + prefetch // 4):
            if pf >= n_ops or pf == p:
                continue
            g2 = order[pf]
            for o in (A[g2], B[g2], C[g2]):
                if o > 1 and loc_v[o] < 0 and loc_a[o] < 0 and mem_slot[o] >= 0:
                    v = alloc_v(empty)
                    loc_v[o] = v
                    issue_load(o, v)
                    heapq.heappush(vheap, (-next_use(o), o))
                    stats["prefetched"] += 1
                    if g2 >= n_eval:
                        stats["loads_for_check"] += 1
        g = order[p]
        ops = (A[g], B[g], C[g])
        pin = set(o for o in ops if o > 1)
        for o in pin:
            if loc_v[o] < 0:
                v = alloc_v(pin)
                loc_v[o] = v
                if loc_a[o] >= 0:
                    emit(("ar", v, loc_a[o]))
                    stats["agpr_reads"] += 1
                    free_a.append(loc_a[o])
                    loc_a[o] = -1
                elif mem_slot[o] >= 0:
                    issue_load(o, v)
                    stats["late_loads"] += 1
                    if g >= n_eval:
                        stats["loads_for_check"] += 1
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
        stored = bool(is_signal[g])
        nu = next_use(g)
        d = alloc_v(pin)
        emit(("g", d, srcs[0], srcs[1], srcs[2], TT[g]))
        stats["gates"] += 1
        loc_v[g] = d
        if stored:
            issue_store("st", d, g)
            stats["stores"] += 1
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
            free_v.append(d)
            loc_v[g] = -1
        else:
            heapq.heappush(vheap, (-nu, g))
    if held_viol[0] >= 0:
        emit(("acc", 2, held_viol[0]))
    if const_assert:
        emit(("accc", 1))
    if const_viol:
        emit(("accc", 2))

    jp = JitProgram()
    jp.ir = ir
    jp.n_slots = (next_slot + 15) // 16 * 16
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


def to_asm(jp: JitProgram) -> str:
    """gfx950 assembly of the program.  Kernel arguments: bit table, fallback masks, R1CS flags (three pointers); one
    wave (workgroup of 64) per chunk of 2 048 instances."""
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
    st_page = -1
    pages = {}                   # page -> sgpr
    lru = []                     # pages, most recent last
    free_s = list(range(20 + N_PAGE_SGPRS - 1, 19, -1))

    def load_page(pg):
        s = pages.get(pg)
        if s is not None:
            if lru[-1] != pg:
                lru.remove(pg)
                lru.append(pg)
            return s
        if free_s:
            s = free_s.pop()
        else:
            old = lru.pop(0)
            s = pages.pop(old)
        pages[pg] = s
        lru.append(pg)
        add("  s_mov_b32 s%d, 0x%x\n" % (s, pg * PAGE))
        return s

    def opnd(x):
        return "0" if x == -1 else "-1" if x == -2 else "v%d" % x

    for ins in jp.ir:
        k = ins[0]
        if k == "g":
            add(gate_asm(ins))
        elif k == "st" or k == "sta":
            off = ins[2] * ROW_BYTES
            pg = off // PAGE
            if pg != st_page:
                add("  s_mov_b32 s16, 0x%x\n" % (pg * PAGE))
                st_page = pg
            add("  buffer_store_dword %s%d, v0, s[12:15], s16 offen offset:%d nt\n" % ("v" if k == "st" else "a", ins[1], off % PAGE))
        elif k == "ld":
            off = ins[2] * ROW_BYTES
            s = load_page(off // PAGE)
            add("  buffer_load_dword v%d, v0, s[12:15], s%d offen offset:%d\n" % (ins[1], s, off % PAGE))
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
        else:
            raise ValueError(k)
    # flags: one dword per lane = 32 instances; fallback masks / R1CS flags are uint64 per group of 64 instances, i.e. dword
    # chunk * 64 + lane of the array
    add("  s_lshl_b32 s17, s2, 8\n  v_add_u32 v0, s17, v0\n"
        "  global_atomic_or v0, v1, s[6:7]\n  global_atomic_or v0, v2, s[8:9]\n"
        "  s_endpgm\n.Lend:\n.size %s, .Lend-%s\n" % (KERNEL_NAME, KERNEL_NAME))
    add(".rodata\n.p2align 6\n.amdhsa_kernel %s\n"
        "  .amdhsa_user_sgpr_kernarg_segment_ptr 1\n  .amdhsa_system_sgpr_workgroup_id_x 1\n  .amdhsa_system_vgpr_workitem_id 0\n"
        "  .amdhsa_next_free_vgpr %d\n  .amdhsa_accum_offset %d\n  .amdhsa_next_free_sgpr 40\n"
        "  .amdhsa_group_segment_fixed_size 0\n  .amdhsa_private_segment_fixed_size 0\n  .amdhsa_kernarg_size 24\n"
        ".end_amdhsa_kernel\n" % (KERNEL_NAME, jp.n_vgpr + jp.n_agpr, jp.n_vgpr))
    add(".amdgpu_metadata\n---\namdhsa.version: [1, 2]\namdhsa.kernels:\n  - .name: %s\n    .symbol: %s.kd\n"
        "    .kernarg_segment_size: 24\n    .group_segment_fixed_size: 0\n    .private_segment_fixed_size: 0\n"
        "    .kernarg_segment_align: 8\n    .wavefront_size: 64\n    .sgpr_count: 40\n    .vgpr_count: %d\n    .agpr_count: %d\n"
        "    .max_flat_workgroup_size: 64\n    .args:\n"
        "      - {.size: 8, .offset: 0, .value_kind: global_buffer, .address_space: global}\n"
        "      - {.size: 8, .offset: 8, .value_kind: global_buffer, .address_space: global}\n"
        "      - {.size: 8, .offset: 16, .value_kind: global_buffer, .address_space: global}\n"
        "...\n.end_amdgpu_metadata\n" % (KERNEL_NAME, KERNEL_NAME, jp.n_vgpr + jp.n_agpr, jp.n_agpr))
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
