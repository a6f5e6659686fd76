"""hip_elements bit-plane lowering, part 1: witness code over provably boolean values -> a network of 3-input gates.

Why: bit-level circomlib circuits (SHA-256, Num2Bits-heavy gadgets) spend one 254-bit field element per BIT: the
reference stores a 40-byte FrElement per signal and runs its short-int paths (generic/fr.cpp:416-439, 696-701,
900-917); the round-1 schedule stored 32 B per bit per instance and was latency/traffic bound.  On a 64-lane
machine the natural layout for a boolean signal is ONE BIT PER INSTANCE: the values of a signal for 64 instances
are one 64-bit mask, a 3-input boolean function of three signals is ~22 VALU instructions for 64 gates x 64
instances when lanes hold 64 different gates ("bit-plane" evaluation, csrc/cw_bits.hip).

This pass abstract-interprets the flat witness code (the straight-line `Fr_*` call sequence the reference emits,
compute_bucket.rs:315-341) under the assumption "the main inputs are 0/1" with integer-exact domains:

  int         a compile-time constant (signed representative of the field element)
  BoolFn      a boolean function (truth table) of <= 3 materialised bits             -> becomes a gate when needed
  Poly        an integer-valued function of <= KMAX bits (exact enumeration: `a*(1-2b-2c+4mid)+b+c-2mid`)
  Lin         sum(coef_i * bit_i) + c0 with arbitrary many terms (`lin += in[j][k] * 2^k` of BinSum)
  BV          a non-negative integer as a vector of BoolFn bits (values of `>>`, `&`, `|`, `^`, word arithmetic)

Field semantics are respected because every domain carries exact integer bounds far below q/2: `+ - *` agree with the
integers, `>> & | ^` and `(x >> k) & 1` are only taken on values proven non-negative (Fr_shr/Fr_band work on the
canonical residue, generic/fr.cpp:1799-2307), and a sum is turned into bits by a carry-save tree + parallel-prefix
adder whose output is exactly the binary expansion of the integer.  `===` checks (assert_bucket.rs:70-89) are
discharged symbolically where possible (out*(out-1) === 0 for a bit; lin === sum(out_k 2^k) for the bits of lin) —
the R1CS check kernel still verifies every constraint on the generated witness — and become assertion gates
otherwise.  Anything the domains cannot express makes the pass give up (`None`): the circuit then runs on the
wide (256-bit) schedule exactly as before.

The assumption is CHECKED at run time: an instance whose inputs are not all 0/1, or that trips an assertion gate, is
flagged and re-evaluated by the wide schedule (csrc/cw_host.cpp), so results are bit-exact for every input.
"""
from __future__ import annotations

import numpy as np

from .. import opcodes as O

K_SIG, K_TMP, K_CONST, K_NONE = O.K_SIG, O.K_TMP, O.K_CONST, O.K_NONE
KMAX = 6                      # leaves of a Poly
BOUND = 1 << 200              # magnitudes stay far below q/2 (q >= 2^225)


class Unsupported(Exception):
    pass


class _Wide:
    """a temporary outside the domains (only an error if it reaches a signal or a run-time check)"""
    __slots__ = ()


_WIDE = _Wide()


# ---- truth-table helpers ------------------------------------------------------------------------------------
# A BoolFn is (leaves, tt): leaves = ascending tuple of node ids (len <= 3), tt = 2^len bits, bit m = value for the
# assignment whose bit j gives leaves[j].
_FULL = (1, 3, 15, 255)


def _tt_expand(leaves, tt, union):
    """re-express tt over the sorted superset `union`"""
    if leaves == union:
        return tt
    pos = [union.index(x) for x in leaves]
    out = 0
    for m in range(1 << len(union)):
        k = 0
        for j, p in enumerate(pos):
            k |= ((m >> p) & 1) << j
        out |= ((tt >> k) & 1) << m
    return out


def _tt_reduce(leaves, tt):
    """drop leaves the function does not depend on"""
    n = len(leaves)
    j = 0
    while j < n:
        lo = hi = 0
        k = 0
        for m in range(1 << n):
            if not (m >> j) & 1:
                lo |= ((tt >> m) & 1) << k
                hi |= ((tt >> (m | (1 << j))) & 1) << k
                k += 1
        if lo == hi:
            leaves = leaves[:j] + leaves[j + 1:]
            tt = lo
            n -= 1
        else:
            j += 1
    return leaves, tt


class Poly:
    """integer-valued function of <= KMAX bits: vals[m] for assignment m (bit j of m = leaves[j])"""
    __slots__ = ("leaves", "vals")

    def __init__(self, leaves, vals):
        self.leaves = leaves
        self.vals = vals


class Lin:
    """c0 + sum coef * node"""
    __slots__ = ("t", "c0", "bv", "sig", "own")

    def __init__(self, t, c0):
        self.t = t
        self.c0 = c0
        self.bv = None
        self.sig = None
        self.own = None                   # the temp that created this object (it may be updated in place by the
                                          # single consumer of that temp: `lin += ...` chains)

    def bounds(self):
        lo = hi = self.c0
        for c in self.t.values():
            if c < 0:
                lo += c
            else:
                hi += c
        return lo, hi


class BV:
    """non-negative integer, bits[k] = BoolFn; `src` = signature of the Lin it is the binary expansion of"""
    __slots__ = ("bits", "src")

    def __init__(self, bits, src=None):
        self.bits = bits
        self.src = src


class BitNet:
    """Result: the gate network.  Node 0 = constant 0, node 1 = constant 1, then the main inputs, then gates."""

    def __init__(self):
        self.tt = [0x00, 0xFF]            # 8-bit table over (a, b, c); inputs carry 0x100
        self.a = [0, 0]
        self.b = [0, 0]
        self.c = [0, 0]
        self.level = [0, 0]
        self.input_node = {}              # main input signal -> node
        self.marks = {}                   # row of the flat code -> number of nodes when the analysis reached it (bitblast(marks=))
        self.port_src = {}                # PORT node -> the node whose value it carries (bitblast(ports=)): the input signals of the
                                          # instances of a repeated template.  The analysis treats a port like a main input -
                                          # nothing is folded or shared through it - and the emitted code reads it from the ROW of
                                          # its source (the reference's template body reads signalValues[mySignalStart + i] the
                                          # same way, template.rs:297-304), so every instance has the same gate network
        self.sig_node = None              # signal -> node (every signal is a bit in bit mode)
        self.asserts = []                 # nodes that must be 0 (violation functions of unproved `===`)
        self.stats = {}

    def __len__(self):
        return len(self.tt)


class _Blaster:
    def __init__(self, fc, ports=None, marks=None):
        self.fc = fc
        # ports: [(first signal, count), ...] per instance of a repeated template - its input signals (ascending, disjoint)
        self.port_group = None
        if ports:
            pg = np.full(fc.n_signals, -1, dtype=np.int64)
            for gi, rngs in enumerate(ports):
                for s0, n_ in rngs:
                    pg[s0:s0 + n_] = gi
            self.port_group = pg.tolist()
            self.port_ranges = ports
            self.port_node = {}           # signal -> its port node (all ports of an instance are created together, in signal order)
        self.marks = set(int(m) for m in marks) if marks is not None else None    # rows of the flat code: node count when reached
        q = fc.fp.q
        half = q >> 1
        self.consts = [c - q if c > half else c for c in fc.constants]
        self.net = BitNet()
        self.cse = {}
        self.bit_of = {}                  # node -> (bv id, k) for bits produced by expanding a Lin
        self.bv_src = {}                  # bv id -> (signature of the Lin, number of bits)
        self.lin_cache = {}               # signature -> BV
        self.n_proved = 0
        self.n_gates = 0
        # arrival levels RELATIVE to the last mark (the start of a repeated template's instance): the carry-save trees order
        # their operands by arrival; with absolute levels an instance fed by the previous instance's outputs would build other
        # adders than the first one, which is fed by inputs and constants (nodes older than `floor` count as level 0)
        self.floor = 0
        self.rlevel = [0, 0]

    # ---- nodes ---------------------------------------------------------------------------------------------
    def new_input(self, sig):
        n = self.net
        nid = len(n.tt)
        n.tt.append(0x100); n.a.append(0); n.b.append(0); n.c.append(0); n.level.append(0)
        self.rlevel.append(0)
        n.input_node[sig] = nid
        return nid

    def port(self, sig, src):
        """the port node of input signal `sig` of a repeated template's instance, now carrying the value of node `src`.  The ports
        of one instance are created TOGETHER when the first of them is assigned, in signal order: their ids - which order the
        operands of every gate - then relate to each other and to the instance's gates the same way in every instance, whatever
        order the parent wires them in and whatever it wires them to"""
        n = self.net
        nid = self.port_node.get(sig)
        if nid is None:
            for s0, cnt in self.port_ranges[self.port_group[sig]]:
                for s_ in range(s0, s0 + cnt):
                    self.port_node[s_] = len(n.tt)
                    n.tt.append(0x100); n.a.append(0); n.b.append(0); n.c.append(0); n.level.append(0)
                    self.rlevel.append(0)
            nid = self.port_node[sig]
        if nid in n.port_src:
            raise Unsupported("an input signal of a component is assigned twice")
        n.port_src[nid] = src
        return nid

    def gate(self, leaves, tt):
        """materialise BoolFn (leaves, tt) -> node id (constants and plain leaves need no gate)"""
        leaves, tt = _tt_reduce(leaves, tt)
        nl = len(leaves)
        if nl == 0:
            return 1 if tt & 1 else 0
        if nl == 1 and tt == 2:
            return leaves[0]
        # pad to three operands (unused = node 0, table replicated)
        if nl == 1:
            t8 = (0xAA if tt == 2 else 0x55)
            key = (t8, leaves[0], 0, 0)
        elif nl == 2:
            t8 = tt | (tt << 4)
            key = (t8, leaves[0], leaves[1], 0)
        else:
            key = (tt, leaves[0], leaves[1], leaves[2])
        nid = self.cse.get(key)
        if nid is None:
            n = self.net
            nid = len(n.tt)
            n.tt.append(key[0]); n.a.append(key[1]); n.b.append(key[2]); n.c.append(key[3])
            lv = n.level
            n.level.append(1 + max(lv[key[1]], lv[key[2]], lv[key[3]]))
            rl, fl = self.rlevel, self.floor
            rl.append(1 + max(rl[key[1]] if key[1] >= fl else 0, rl[key[2]] if key[2] >= fl else 0, rl[key[3]] if key[3] >= fl else 0))
            self.cse[key] = nid
            self.n_gates += 1
        return nid

    def node_of(self, f):
        return self.gate(f[0], f[1])

    # ---- BoolFn algebra ------------------------------------------------------------------------------------
    def bf_norm(self, f):
        """BoolFn whose support exceeds 3 never exists; reduce constants/identities"""
        leaves, tt = _tt_reduce(f[0], f[1])
        return (leaves, tt)

    def bf2(self, f, g, op):
        """op: 4-bit table over (f, g): bit (f + 2g)"""
        lf, lg = f[0], g[0]
        if lf == lg:
            union = lf
        else:
            union = tuple(sorted(set(lf) | set(lg)))
            if len(union) > 3:
                # materialise the operand with the larger support first, then the other if still needed
                if len(lf) >= len(lg):
                    f = ((self.node_of(f),), 2)
                else:
                    g = ((self.node_of(g),), 2)
                lf, lg = f[0], g[0]
                union = tuple(sorted(set(lf) | set(lg)))
                if len(union) > 3:
                    f = ((self.node_of(f),), 2)
                    g = ((self.node_of(g),), 2)
                    lf, lg = f[0], g[0]
                    union = tuple(sorted(set(lf) | set(lg)))
        tf = _tt_expand(lf, f[1], union)
        tg = _tt_expand(lg, g[1], union)
        full = _FULL[len(union)]
        nf, ng = full ^ tf, full ^ tg
        r = 0
        if op & 1:
            r |= nf & ng
        if op & 2:
            r |= tf & ng
        if op & 4:
            r |= nf & tg
        if op & 8:
            r |= tf & tg
        return _tt_reduce(union, r)

    def bf3(self, x, y, z, tt8):
        """gate over three MATERIALISED nodes (full-adder cells)"""
        leaves = tuple(sorted({x, y, z}))
        if len(leaves) < 3 or 0 in leaves or 1 in leaves:
            # repeated / constant operands: rebuild the table over the distinct non-constant leaves
            real = tuple(v for v in leaves if v > 1)
            tt = 0
            for m in range(1 << len(real)):
                val = {0: 0, 1: 1}
                for j, v in enumerate(real):
                    val[v] = (m >> j) & 1
                k = val[x] | (val[y] << 1) | (val[z] << 2)
                tt |= ((tt8 >> k) & 1) << m
            return self.gate(real, tt)
        # permute tt8 (given over x,y,z order) to sorted order
        order = (x, y, z)
        tt = 0
        for m in range(8):
            val = {leaves[0]: m & 1, leaves[1]: (m >> 1) & 1, leaves[2]: (m >> 2) & 1}
            k = val[order[0]] | (val[order[1]] << 1) | (val[order[2]] << 2)
            tt |= ((tt8 >> k) & 1) << m
        return self.gate(leaves, tt)

    # ---- conversions -----------------------------------------------------------------------------------------
    @staticmethod
    def is_bf(v):
        return type(v) is tuple

    def to_poly(self, v):
        """-> Poly or None"""
        t = type(v)
        if t is Poly:
            return v
        if t is int:
            return Poly((), (v,))
        if t is tuple:
            return Poly(v[0], tuple((v[1] >> m) & 1 for m in range(1 << len(v[0]))))
        if t is Lin:
            if len(v.t) > KMAX:
                return None
            leaves = tuple(sorted(v.t))
            cs = [v.t[x] for x in leaves]
            vals = []
            for m in range(1 << len(leaves)):
                s = v.c0
                for j, c in enumerate(cs):
                    if (m >> j) & 1:
                        s += c
                vals.append(s)
            return Poly(leaves, tuple(vals))
        if t is BV:
            leaves = set()
            for f in v.bits:
                leaves.update(f[0])
            if len(leaves) > KMAX:
                return None
            leaves = tuple(sorted(leaves))
            vals = [0] * (1 << len(leaves))
            for k, f in enumerate(v.bits):
                tf = _tt_expand(f[0], f[1], leaves) if len(leaves) <= 3 else self._tt_expand_big(f, leaves)
                for m in range(len(vals)):
                    if (tf >> m) & 1:
                        vals[m] += 1 << k
            return Poly(leaves, tuple(vals))
        return None

    @staticmethod
    def _tt_expand_big(f, union):
        pos = [union.index(x) for x in f[0]]
        out = 0
        for m in range(1 << len(union)):
            k = 0
            for j, p in enumerate(pos):
                k |= ((m >> p) & 1) << j
            out |= ((f[1] >> k) & 1) << m
        return out

    def poly_simplify(self, p):
        """drop leaves the values do not depend on; eliminate leaves that are gates over other leaves of the poly
        (`mid = b*c` inside `a*(1-2b-2c+4mid)+...`): only consistent assignments are kept"""
        leaves, vals = p.leaves, p.vals
        net = self.net
        changed = True
        while changed and len(leaves) > 0:
            changed = False
            n = len(leaves)
            # independent leaves
            for j in range(n):
                if all(vals[m] == vals[m | (1 << j)] for m in range(1 << n) if not (m >> j) & 1):
                    vals = tuple(vals[m] for m in range(1 << n) if not (m >> j) & 1)
                    leaves = leaves[:j] + leaves[j + 1:]
                    changed = True
                    break
            if changed:
                continue
            if n <= 3:
                break
            # a leaf defined by a gate over other leaves (or constants)
            for j in range(n - 1, -1, -1):
                x = leaves[j]
                if net.tt[x] > 0xFF:
                    continue
                ops = (net.a[x], net.b[x], net.c[x])
                if all(o <= 1 or (o in leaves and o != x) for o in ops):
                    idx = [(-1 if o <= 1 else leaves.index(o)) for o in ops]
                    t8 = net.tt[x]
                    nv = []
                    for m in range(1 << n):
                        if (m >> j) & 1:
                            continue
                        k = 0
                        for jj, (o, ix) in enumerate(zip(ops, idx)):
                            bit = o if ix < 0 else (m >> ix) & 1
                            k |= bit << jj
                        xv = (t8 >> k) & 1
                        nv.append(vals[m | (xv << j)])
                    vals = tuple(nv)
                    leaves = leaves[:j] + leaves[j + 1:]
                    changed = True
                    break
        return Poly(leaves, vals)

    def poly_result(self, p):
        """canonical form of a Poly result: int if constant, BoolFn if boolean over <= 3 leaves"""
        p = self.poly_simplify(p)
        if not p.leaves:
            return p.vals[0]
        if len(p.leaves) <= 3 and all(v == 0 or v == 1 for v in p.vals):
            tt = 0
            for m, v in enumerate(p.vals):
                tt |= v << m
            return (p.leaves, tt)
        return p

    def to_lin(self, v):
        t = type(v)
        if t is Lin:
            return v
        if t is int:
            return Lin({}, v)
        if t is tuple:
            n = self.node_of(v)
            if n <= 1:
                return Lin({}, n)
            return Lin({n: 1}, 0)
        if t is Poly:
            # affine in its leaves?  (b - c, 1 - 2b - 2c + 4mid, ...)
            n = len(v.leaves)
            c0 = v.vals[0]
            cs = [v.vals[1 << j] - c0 for j in range(n)]
            ok = True
            for m in range(1 << n):
                s = c0
                for j in range(n):
                    if (m >> j) & 1:
                        s += cs[j]
                if s != v.vals[m]:
                    ok = False
                    break
            if ok:
                return Lin({x: c for x, c in zip(v.leaves, cs) if c}, c0)
            lo = min(v.vals)
            bv = self.poly_to_bv(Poly(v.leaves, tuple(x - lo for x in v.vals)))
            L = self.to_lin(bv)
            return Lin(dict(L.t), L.c0 + lo)
        if t is BV:
            tm = {}
            c0 = 0
            for k, f in enumerate(v.bits):
                n = self.node_of(f)
                if n == 1:
                    c0 += 1 << k
                elif n > 1:
                    tm[n] = tm.get(n, 0) + (1 << k)
            return Lin(tm, c0)
        raise Unsupported("to_lin")

    def poly_to_bv(self, p):
        """Poly with non-negative values -> BV (one BoolFn per output bit; > 3 leaves: Shannon expansion)"""
        hi = max(p.vals)
        if min(p.vals) < 0:
            raise Unsupported("negative value used as an unsigned integer")
        nb = max(hi.bit_length(), 1)
        bits = []
        for k in range(nb):
            tt = 0
            for m, v in enumerate(p.vals):
                tt |= ((v >> k) & 1) << m
            bits.append(self.shannon(p.leaves, tt))
        return BV(bits)

    def shannon(self, leaves, tt):
        """boolean function of any number of leaves -> BoolFn (<= 3 leaves), building mux gates as needed"""
        n = len(leaves)
        if n <= 3:
            return _tt_reduce(leaves, tt)
        # reduce support first
        for j in range(n):
            lo = hi = 0
            k = 0
            for m in range(1 << n):
                if not (m >> j) & 1:
                    lo |= ((tt >> m) & 1) << k
                    hi |= ((tt >> (m | (1 << j))) & 1) << k
                    k += 1
            if lo == hi:
                return self.shannon(leaves[:j] + leaves[j + 1:], lo)
        # split on the last leaf
        j = n - 1
        half = 1 << j
        lo = tt & ((1 << half) - 1)
        hi = tt >> half
        f0 = self.shannon(leaves[:j], lo)
        f1 = self.shannon(leaves[:j], hi)
        n0, n1 = self.node_of(f0), self.node_of(f1)
        # mux(s, n1, n0) = s ? n1 : n0 over (n0, n1, s): bit index = n0 + 2 n1 + 4 s -> 0xCA
        g = self.bf3(n0, n1, leaves[j], 0xCA)
        return ((g,), 2) if g > 1 else ((), g)

    def to_bv(self, v):
        t = type(v)
        if t is BV:
            return v
        if t is int:
            if v < 0:
                raise Unsupported("negative constant used as an unsigned integer")
            return BV([((), (v >> k) & 1) for k in range(max(v.bit_length(), 1))])
        if t is tuple:
            return BV([v])
        if t is Poly:
            return self.poly_to_bv(self.poly_simplify(v))
        if t is Lin:
            return self.lin_to_bv(v)
        raise Unsupported("to_bv")

    # ---- Lin -> BV: carry-save tree + parallel-prefix adder --------------------------------------------------
    def lin_sig(self, L):
        if L.sig is None:
            L.sig = (L.c0, tuple(sorted(L.t.items())))
        return L.sig

    def lin_to_bv(self, L):
        if L.bv is not None:
            return L.bv
        sig = self.lin_sig(L)
        got = self.lin_cache.get(sig)
        if got is not None:
            L.bv = got
            return got
        lo, hi = L.bounds()
        if lo < 0:
            raise Unsupported("a possibly negative sum is used as an unsigned integer")
        if hi >= BOUND:
            raise Unsupported("sum too large")
        nb = max(hi.bit_length(), 1)
        net = self.net
        cols = [[] for _ in range(nb)]
        c0 = L.c0
        for node, coef in sorted(L.t.items()):
            if coef < 0:                           # coef*x = coef + |coef| * (1 - x)
                c0 += coef
                lit = self.gate((node,), 1)        # NOT x
                coef = -coef
            else:
                lit = node
            k = 0
            while coef and k < nb:
                if coef & 1:
                    cols[k].append(lit)
                coef >>= 1
                k += 1
        c0 %= (1 << nb)                            # the true value is in [0, 2^nb): arithmetic mod 2^nb is exact
        for k in range(nb):
            if (c0 >> k) & 1:
                cols[k].append(1)
        lv, fl = self.rlevel, self.floor
        # Wallace rounds: every column with >= 3 entries is cut into triples (earliest arrivals first)
        while max(len(c) for c in cols) > 2:
            nxt = [[] for _ in range(nb)]
            for k in range(nb):
                col = sorted(cols[k], key=lambda x: lv[x] if x >= fl else 0)
                i = 0
                while len(col) - i >= 3:
                    x, y, z = col[i], col[i + 1], col[i + 2]
                    i += 3
                    s = self.bf3(x, y, z, 0x96)
                    if s:
                        nxt[k].append(s)
                    if k + 1 < nb:
                        cy = self.bf3(x, y, z, 0xE8)
                        if cy:
                            nxt[k + 1].append(cy)
                nxt[k].extend(col[i:])
            cols = nxt
        # two rows -> generate/propagate, Kogge-Stone prefix, sum
        g = [0] * nb
        p = [0] * nb
        for k in range(nb):
            col = cols[k]
            if len(col) == 2:
                g[k] = self.bf3(col[0], col[1], 0, 0x88)          # x & y
                p[k] = self.bf3(col[0], col[1], 0, 0x66)          # x ^ y
            elif len(col) == 1:
                p[k] = col[0]
        G, P = list(g), list(p)
        d = 1
        while d < nb:
            nG, nP = list(G), list(P)
            for k in range(d, nb):
                if P[k] == 0:
                    continue                                       # (G, 0) absorbs nothing
                if G[k - d] != 0:
                    nG[k] = self.bf3(G[k], P[k], G[k - d], 0xEA)   # G | (P & G')   over (G, P, G'): idx = G+2P+4G'
                if k >= 2 * d:                                     # later rounds only combine positions >= 2d
                    nP[k] = self.bf3(P[k], P[k - d], 0, 0x88) if P[k - d] != 0 else 0
            G, P = nG, nP
            d <<= 1
        bits = []
        for k in range(nb):
            cin = G[k - 1] if k else 0
            s = self.bf3(p[k], cin, 0, 0x66)
            bits.append(((s,), 2) if s > 1 else ((), s))
        bv = BV(bits)
        bid = len(self.bv_src)
        self.bv_src[bid] = (sig, nb)
        bv.src = bid
        for k, f in enumerate(bits):
            if f[0]:
                self.bit_of.setdefault(f[0][0], (bid, k))
        L.bv = bv
        self.lin_cache[sig] = bv
        return bv

    # ---- arithmetic ------------------------------------------------------------------------------------------
    def add(self, x, y, sign, steal):
        """x + sign*y"""
        tx, ty = type(x), type(y)
        if tx is int and ty is int:
            return x + sign * y
        if tx is not Lin and ty is not Lin:
            px, py = self.to_poly(x), self.to_poly(y)
            if px is not None and py is not None:
                r = self.poly_bin(px, py, (lambda a, b: a + b) if sign > 0 else (lambda a, b: a - b))
                if r is not None:
                    return r
        lx = self.to_lin(x)
        ly = self.to_lin(y)
        if steal and lx is x:
            t = lx.t
        else:
            t = dict(lx.t)
        for n, c in ly.t.items():
            v = t.get(n, 0) + sign * c
            if v:
                t[n] = v
            else:
                t.pop(n, None)
        r = Lin(t, lx.c0 + sign * ly.c0)
        lo, hi = r.bounds() if len(t) < 64 else (0, 0)
        if max(abs(lo), abs(hi)) >= BOUND:
            raise Unsupported("sum out of range")
        return r

    def poly_bin(self, px, py, fn):
        if px.leaves == py.leaves:
            leaves = px.leaves
            vals = tuple(fn(a, b) for a, b in zip(px.vals, py.vals))
        else:
            leaves = tuple(sorted(set(px.leaves) | set(py.leaves)))
            if len(leaves) > KMAX:
                return None
            ix = [leaves.index(v) for v in px.leaves]
            iy = [leaves.index(v) for v in py.leaves]
            vals = []
            for m in range(1 << len(leaves)):
                kx = ky = 0
                for j, p_ in enumerate(ix):
                    kx |= ((m >> p_) & 1) << j
                for j, p_ in enumerate(iy):
                    ky |= ((m >> p_) & 1) << j
                vals.append(fn(px.vals[kx], py.vals[ky]))
            vals = tuple(vals)
        if max(abs(min(vals)), abs(max(vals))) >= BOUND:
            raise Unsupported("value out of range")
        return self.poly_result(Poly(leaves, vals))

    def mul(self, x, y):
        tx, ty = type(x), type(y)
        if tx is int and ty is int:
            return x * y
        if ty is int:
            x, y, tx, ty = y, x, ty, tx
        if tx is int:
            if x == 0:
                return 0
            if x == 1:
                return y
            if ty is tuple or ty is Poly:
                p = self.to_poly(y)
                return self.poly_result(Poly(p.leaves, tuple(v * x for v in p.vals)))
            if ty is BV and x > 0 and x & (x - 1) == 0:
                return BV([((), 0)] * (x.bit_length() - 1) + list(y.bits))
            L = self.to_lin(y)
            r = Lin({n: c * x for n, c in L.t.items()}, L.c0 * x)
            return r
        px, py = self.to_poly(x), self.to_poly(y)
        if px is not None and py is not None:
            r = self.poly_bin(px, py, lambda a, b: a * b)
            if r is not None:
                return r
        # bit * vector: AND every bit
        if tx is tuple or ty is tuple:
            bit, oth = (x, y) if tx is tuple else (y, x)
            bv = self.to_bv(oth)
            return BV([self.bf2(bit, f, 8) for f in bv.bits])
        raise Unsupported("product of two wide run-time values")

    def shift_amount(self, y):
        if type(y) is not int or y < 0 or y > 250:
            raise Unsupported("shift by a run-time or huge amount")
        return y

    def bitwise(self, x, y, op4):
        bx, by = self.to_bv(x), self.to_bv(y)
        n = max(len(bx.bits), len(by.bits))
        zero = ((), 0)
        out = []
        for k in range(n):
            f = bx.bits[k] if k < len(bx.bits) else zero
            g = by.bits[k] if k < len(by.bits) else zero
            out.append(self.bf2(f, g, op4))
        while len(out) > 1 and out[-1] == zero:
            out.pop()
        return BV(out)

    # ---- assertions ------------------------------------------------------------------------------------------
    def recompose(self, L):
        """if the terms of L are exactly the bits of one expanded sum with weights 2^k, return that sum's signature"""
        if not L.t:
            return None
        first = next(iter(L.t))
        ent = self.bit_of.get(first)
        if ent is None:
            return None
        bid = ent[0]
        sig, nb = self.bv_src[bid]
        bv = self.lin_cache[sig]
        tm = {}
        c0 = 0
        for k, f in enumerate(bv.bits):
            if f[0]:
                tm[f[0][0]] = tm.get(f[0][0], 0) + (1 << k)
            elif f[1] & 1:
                c0 += 1 << k
        if tm == L.t and c0 == L.c0:
            return sig
        return None

    def assert_eq(self, x, y):
        d = self.add(x, y, -1, False)
        td = type(d)
        if td is int:
            if d == 0:
                self.n_proved += 1
                return
            raise Unsupported("assertion that always fails")
        if td is tuple:
            self.net.asserts.append(self.node_of(d))
            return
        if td is Poly:
            # violated iff value != 0
            tt = 0
            for m, v in enumerate(d.vals):
                if v != 0:
                    tt |= 1 << m
            if tt == 0:
                self.n_proved += 1
                return
            self.net.asserts.append(self.node_of(self.shannon(d.leaves, tt)))
            return
        # Lin: sum === its own binary expansion?
        lx, ly = self.to_lin(x), self.to_lin(y)
        for u, v in ((lx, ly), (ly, lx)):
            sig = self.recompose(u)
            if sig is not None and sig == self.lin_sig(v):
                self.n_proved += 1
                return
        if not d.t and d.c0 == 0:
            self.n_proved += 1
            return
        # general case: compare the binary expansions of both sides (both must be non-negative)
        bx, by = self.to_bv(lx), self.to_bv(ly)
        neq = self.bitwise(bx, by, 6)
        acc = ((), 0)
        for f in neq.bits:
            acc = self.bf2(acc, f, 14)
        n = self.node_of(acc)
        if n == 0:
            self.n_proved += 1
        elif n == 1:
            raise Unsupported("assertion that always fails")
        else:
            self.net.asserts.append(n)

    def step(self, o, x, y, steal):
        """one flat operator on abstract values (temporaries)"""
        if o == O.COPY:
            return x
        if o == O.NEG:
            return self.mul(-1, x)
        if y is None:
            raise Unsupported("operator %s" % O.NAMES[o])
        if o == O.ADD or o == O.SUB:
            r = self.add(x, y, 1 if o == O.ADD else -1, steal)
        elif o == O.MUL:
            r = self.mul(x, y)
        elif o == O.SHL:
            r = self.mul(1 << self.shift_amount(y), x)
        elif o == O.SHR:
            k = self.shift_amount(y)
            b = self.to_bv(x)
            r = BV(b.bits[k:] if k < len(b.bits) else [((), 0)])
        elif o == O.BAND:
            if type(y) is int and y >= 0 and type(x) is not int:
                b = self.to_bv(x)
                bits = [f for k, f in enumerate(b.bits) if k < y.bit_length()]
                r = BV([f if (y >> k) & 1 else ((), 0) for k, f in enumerate(bits)] or [((), 0)])
            elif type(x) is int and x >= 0 and type(y) is not int:
                b = self.to_bv(y)
                bits = [f for k, f in enumerate(b.bits) if k < x.bit_length()]
                r = BV([f if (x >> k) & 1 else ((), 0) for k, f in enumerate(bits)] or [((), 0)])
            else:
                r = self.bitwise(x, y, 8)
        elif o == O.BOR:
            r = self.bitwise(x, y, 14)
        elif o == O.BXOR:
            r = self.bitwise(x, y, 6)
        else:
            raise Unsupported("operator %s" % O.NAMES[o])
        if type(r) is BV and len(r.bits) == 1:
            r = r.bits[0]
            if not r[0]:
                r = r[1] & 1
        return r

    # ---- main loop ------------------------------------------------------------------------------------------------
    def run(self):
        fc = self.fc
        code = fc.code
        op = code["op"].tolist()
        dk = code["dk"].tolist(); dv = code["dv"].tolist()
        ak = code["ak"].tolist(); av = code["av"].tolist()
        bk = code["bk"].tolist(); bv_ = code["bv"].tolist()
        n = len(op)
        uses = np.zeros(max(fc.n_temps, 1), dtype=np.int32)
        for kk, vv in (("ak", "av"), ("bk", "bv"), ("ck", "cv")):
            m = code[kk] == K_TMP
            np.add.at(uses, code[vv][m], 1)
        uses = uses.tolist()
        consts = self.consts
        port_group = self.port_group
        sig = [None] * fc.n_signals
        sig[0] = ((), 1)
        for k in range(fc.n_main_inputs):
            s = fc.main_input_start + k
            sig[s] = ((self.new_input(s),), 2)
        tmp = [None] * max(fc.n_temps, 1)

        def rd(k, v):
            if k == K_SIG:
                x = sig[v]
                if x is None:
                    raise Unsupported("signal read before it is assigned")
                return x
            if k == K_TMP:
                return tmp[v]
            return consts[v]

        ADD, SUB, MUL, NEG, COPY = O.ADD, O.SUB, O.MUL, O.NEG, O.COPY
        SHL, SHR, BAND, BOR, BXOR = O.SHL, O.SHR, O.BAND, O.BOR, O.BXOR
        WIDE = _WIDE
        marks = self.marks
        mark_at = self.net.marks
        for i in range(n):
            if marks is not None and i in marks:
                mark_at[i] = self.floor = len(self.net.tt)
            o = op[i]
            if o == O.RUN:
                continue
            if o == O.CALL:
                raise Unsupported("a function with run-time control flow")
            x = rd(ak[i], av[i])
            # a temporary the domains cannot express (field-sized weights, divisions, ...) only matters if it reaches a
            # signal or a run-time check: dead helper computations (sums that exist for a constraint only) are common
            if dk[i] == K_TMP and o != O.ASSERT_EQ:
                y0 = rd(bk[i], bv_[i]) if bk[i] != K_NONE else None
                if x is WIDE or y0 is WIDE:
                    tmp[dv[i]] = WIDE
                    continue
                try:
                    r = self.step(o, x, y0, ak[i] == K_TMP and uses[av[i]] == 1 and type(x) is Lin and x.own == av[i])
                except Unsupported:
                    r = WIDE
                if type(r) is Lin and r.own is None:
                    r.own = dv[i]
                tmp[dv[i]] = r
                continue
            if x is WIDE or (bk[i] != K_NONE and rd(bk[i], bv_[i]) is WIDE):
                raise Unsupported("a value outside the boolean/small-integer domains reaches a signal or a check")
            if o == COPY:
                r = x
            elif o == O.ASSERT_EQ:
                self.assert_eq(x, rd(bk[i], bv_[i]))
                continue
            else:
                r = self.step(o, x, rd(bk[i], bv_[i]) if bk[i] != K_NONE else None, False)
            if dk[i] == K_SIG:
                # every signal must be a bit in bit mode
                if type(r) is int:
                    if r != 0 and r != 1:
                        raise Unsupported("signal holds a constant that is not a bit")
                    r = ((), r)
                elif type(r) is not tuple:
                    p = self.to_poly(r)
                    r = self.poly_result(p) if p is not None else r
                    if type(r) is int and r in (0, 1):
                        r = ((), r)
                    if type(r) is Poly and all(v == 0 or v == 1 for v in r.vals):
                        # a boolean function of 4..KMAX leaves (constants folded into one side of an xor3 / maj of a short
                        # message): mux tree over 3-input gates
                        tt = 0
                        for m, v in enumerate(r.vals):
                            tt |= v << m
                        r = self.shannon(r.leaves, tt)
                    if type(r) is not tuple:
                        raise Unsupported("signal %d is not provably a bit" % dv[i])
                nid = self.node_of(r)
                if port_group is not None and port_group[dv[i]] >= 0:
                    nid = self.port(dv[i], nid)
                sig[dv[i]] = ((nid,), 2) if nid > 1 else ((), nid)
            elif dk[i] == K_TMP:
                if type(r) is Lin and r.own is None:
                    r.own = dv[i]
                tmp[dv[i]] = r
        net = self.net
        if marks is not None and n in marks:
            mark_at[n] = len(net.tt)
        sn = np.zeros(fc.n_signals, dtype=np.int64)
        for s, v in enumerate(sig):
            if v is None:
                raise Unsupported("signal %d is never assigned" % s)
            sn[s] = v[0][0] if v[0] else (v[1] & 1)
        net.sig_node = sn
        net.stats = {"gates": self.n_gates, "nodes": len(net.tt), "inputs": len(net.input_node),
                     "asserts_proved": self.n_proved, "asserts_left": len(net.asserts),
                     "depth": max(net.level) if net.level else 0}
        return net


def bitblast(fc, ports=None, marks=None):
    """FlatCircuit -> BitNet, or None when the circuit is not (entirely) a boolean computation of 0/1 inputs.
    ports: per instance of a repeated template the ranges [(first signal, count)] of its INPUT signals: whatever is stored into
    one of them - a constant, an input, the previous instance's output - is not propagated: the signal gets a PORT node of its
    own (`BitNet.port_src`: port -> the node it carries).  The code emitter asks for this (bitjit.instance_ports), so that every
    instance has the same gate network whatever its inputs are wired to, and reads the ports from their sources' rows.
    marks: rows of the flat code; `BitNet.marks[row]` = number of nodes when the analysis reached that row (the node range of
    a component's subtree, with FlatCircuit.comp_code_range)."""
    try:
        return _Blaster(fc, ports, marks).run()
    except Unsupported as e:
        bitblast.why = str(e)
        return None


bitblast.why = ""


# ---- reference simulation of a BitNet (test helper; python ints as masks over any number of instances) --------------
def simulate(net: BitNet, input_masks: dict, width: int):
    """input_masks: main input signal -> int mask (bit i = value in instance i).  Returns list of node masks."""
    full = (1 << width) - 1
    n = len(net.tt)
    val = [0] * n
    val[1] = full
    for s, nid in net.input_node.items():
        val[nid] = input_masks[s] & full
    tt, a, b, c = net.tt, net.a, net.b, net.c
    src = getattr(net, "port_src", {})

    def rs(x):                                   # a port reads the node it carries (which may be created after the port)
        while x in src:
            x = src[x]
        return x
    for i in range(2, n):
        t = tt[i]
        if t > 0xFF:
            continue
        A, B, C = val[rs(a[i])], val[rs(b[i])], val[rs(c[i])]
        nA, nB = full ^ A, full ^ B
        lo = ((nA & nB) if t & 1 else 0) | ((A & nB) if t & 2 else 0) | ((nA & B) if t & 4 else 0) | ((A & B) if t & 8 else 0)
        hi = ((nA & nB) if t & 16 else 0) | ((A & nB) if t & 32 else 0) | ((nA & B) if t & 64 else 0) | ((A & B) if t & 128 else 0)
        val[i] = (lo & (full ^ C)) | (hi & C)
    for p_ in src:
        val[p_] = val[rs(p_)]
    return val
