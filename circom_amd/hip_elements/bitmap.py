"""hip_elements bit-plane lowering, part 1b: technology mapping of the gate network onto the kernel's gate PRIMITIVE.

bitblast.py produces a network of arbitrary 3-input gates (8-bit truth tables).  Evaluating a lane-specific truth table
costs the kernel 8 `v_bfe_i32` + 14 `v_bfi_b32` per vrow (round 2).  gfx950 has `v_bitop3_b32` (any 3-input boolean
function, table in the instruction), but a table in the instruction is the same for all 64 lanes — so the kernel
evaluates ONE two-stage primitive whose variant is picked per lane by two mask bits (cw_bits.hip):

    u = K1 ? (a & b) : (a ^ b)            one v_bitop3_b32 per 32 instances, inputs (a, b, K1)
    r = K2 ? (u | c) : (u ^ c)            one v_bitop3_b32 per 32 instances, inputs (u, c, K2)

with a, b, c ring entries or the constants 0 / 1 (two fixed LDS entries).  That covers AND, XOR, XOR3, OR, AND-OR
(Kogge-Stone's g | p & g'), AND-XOR, NOT/XNOR (c = 1) in one primitive - 91 % of the gates of SHA-256 - and everything
else through a small recipe (MAJ = b ^ ((a ^ b) & (b ^ c)): 3 primitives, 2 levels; MUX = y ^ (s & (x ^ y)): 2, 2).
Recipes for all 256 tables are found once by exhaustive search (fewest levels, then fewest primitives); shared
sub-terms (a ^ b of a full adder's sum and carry) are merged by structural hashing.
"""
from __future__ import annotations

import numpy as np

from .bitblast import BitNet

K_AND = 1          # K1: stage 1 is AND (else XOR)
K_OR = 2           # K2: stage 2 is OR (else XOR)


def _prim_eval(p, q, r, k):
    """truth tables (uint8 arrays or ints over the 8 assignments) of the primitive's result"""
    u = (p & q) if (k & K_AND) else (p ^ q)
    return (u | r) if (k & K_OR) else (u ^ r)


_RECIPES = None
_NEG = -99


def recipes():
    """Pareto-optimal recipes of every 3-input function: rec[tt] = list of (dx, dy, dz, cost, tree) where d* = number of
    primitive levels between that operand and the output (-99: the output does not depend on it through this tree), cost =
    primitives (tree count) and tree = 'x' | 'y' | 'z' | 0 | 1 | (k, tree_p, tree_q, tree_r).  All trees of at most 3
    primitives are enumerated (every function has one: checked), so that the mapper can pick, per gate, the tree whose
    output arrives first given WHEN its operands arrive - a full adder's carry is `((x ^ y) & z) | (x & y)`: one level
    behind the late operand z, where the symmetric form needs two."""
    global _RECIPES
    if _RECIPES is not None:
        return _RECIPES
    base = {0xAA: ((0, _NEG, _NEG, 0), 'x'), 0xCC: ((_NEG, 0, _NEG, 0), 'y'), 0xF0: ((_NEG, _NEG, 0, 0), 'z'),
            0x00: ((_NEG, _NEG, _NEG, 0), 0), 0xFF: ((_NEG, _NEG, _NEG, 0), 1)}
    # by_cost[c] = list of (tt, delays(dx,dy,dz), tree)
    by_cost = {0: [(f, d[:3], t) for f, (d, t) in base.items()]}
    pareto = {f: [(d[0], d[1], d[2], 0, t)] for f, (d, t) in base.items()}

    def offer(f, dx, dy, dz, cost, tree):
        lst = pareto.setdefault(f, [])
        for e in lst:
            if e[0] <= dx and e[1] <= dy and e[2] <= dz and e[3] <= cost:
                return False
        lst[:] = [e for e in lst if not (dx <= e[0] and dy <= e[1] and dz <= e[2] and cost <= e[3])]
        lst.append((dx, dy, dz, cost, tree))
        return True

    def combine(cp, cq, cr):
        new = []
        for (fp, dp, tp) in by_cost.get(cp, ()):
            for (fq, dq, tq) in by_cost.get(cq, ()):
                if cp == cq and fq < fp:
                    continue                                   # first stage is commutative
                u_and, u_xor = fp & fq, fp ^ fq
                d2 = (max(dp[0], dq[0]), max(dp[1], dq[1]), max(dp[2], dq[2]))
                for (fr, dr, tr) in by_cost.get(cr, ()):
                    d = tuple((max(d2[i], dr[i]) + 1) if max(d2[i], dr[i]) > _NEG else _NEG for i in range(3))
                    for k in range(4):
                        u = u_and if (k & K_AND) else u_xor
                        f = (u | fr) if (k & K_OR) else (u ^ fr)
                        if f in base:
                            continue
                        c = cp + cq + cr + 1
                        tree = (k, tp, tq, tr)
                        if offer(f, d[0], d[1], d[2], c, tree):
                            new.append((f, d, tree))
        return new

    for total in (1, 2, 3):
        fresh = []
        for cp in range(total):
            for cq in range(total - cp):
                cr = total - 1 - cp - cq
                fresh += combine(cp, cq, cr)
        # keep only entries that survived in the Pareto sets
        alive = {(f, e[4]) for f, lst in pareto.items() for e in lst if e[3] == total}
        seen = set()
        lvl = []
        for f, d, tree in fresh:
            if (f, tree) in alive and (f, tree) not in seen:
                seen.add((f, tree))
                lvl.append((f, d, tree))
        by_cost[total] = lvl
    assert len(pareto) == 256, len(pareto)
    _RECIPES = pareto
    return _RECIPES


class PrimNet:
    """Node 0 = constant 0, node 1 = constant 1, then main inputs (kind -1), then primitives (kind = K1 | K2)."""

    def __init__(self):
        self.kind = [-2, -2]
        self.a = [0, 0]
        self.b = [0, 0]
        self.c = [0, 0]
        self.level = [0, 0]
        self.input_node = {}
        self.sig_node = None
        self.asserts = []
        self.stats = {}

    def __len__(self):
        return len(self.kind)


def map_network(net: BitNet) -> PrimNet:
    rec = recipes()
    out = PrimNet()
    n = len(net.tt)
    new_id = [0] * n
    new_id[1] = 1
    cse = {}
    kind, A, B, C, level = out.kind, out.a, out.b, out.c, out.level

    def prim(k, p, q, r):
        # normal forms: commutative first stage; constant folding of trivial primitives
        if p > q:
            p, q = q, p
        u_zero, u_node = False, None
        if k & K_AND:
            if p == 0:                   # 0 & q
                u_zero = True
            elif p == 1 or p == q:       # 1 & q = q, q & q = q
                u_node = q
        else:
            if p == q:
                u_zero = True
            elif p == 0:
                u_node = q
        if u_zero:                       # stage 1 is the constant 0: result = r (OR or XOR with 0)
            return r
        if u_node is not None:
            # stage 1 passes a node: result = node | r or node ^ r
            if r == 0:
                return u_node
            if k & K_OR:
                if r == 1:
                    return 1
                if r == u_node:
                    return u_node
            elif r == u_node:
                return 0
            # a proper primitive with a constant partner: (0 ^ node) op r
            p, q = 0, u_node
            k = k & K_OR
        key = (k, p, q, r)
        nid = cse.get(key)
        if nid is None:
            nid = len(kind)
            kind.append(k); A.append(p); B.append(q); C.append(r)
            level.append(1 + max(level[p], level[q], level[r]))
            cse[key] = nid
        return nid

    def build(tree, leaves):
        if tree == 'x':
            return leaves[0]
        if tree == 'y':
            return leaves[1]
        if tree == 'z':
            return leaves[2]
        if tree == 0 or tree == 1:
            return tree
        k, tp, tq, tr = tree
        return prim(k, build(tp, leaves), build(tq, leaves), build(tr, leaves))

    choice_cache = {}
    for nid in range(2, n):
        t = net.tt[nid]
        if t == 0x100:
            kind.append(-1); A.append(0); B.append(0); C.append(0); level.append(0)
            new_id[nid] = len(kind) - 1
            continue
        leaves = (new_id[net.a[nid]], new_id[net.b[nid]], new_id[net.c[nid]])
        lx, ly, lz = level[leaves[0]], level[leaves[1]], level[leaves[2]]
        base_l = min(lx, ly, lz)
        ck = (t, lx - base_l, ly - base_l, lz - base_l)
        tree = choice_cache.get(ck)
        if tree is None:
            bestk = None
            for dx, dy, dz, cost, tr in rec[t]:
                arr = max(lx - base_l + dx, ly - base_l + dy, lz - base_l + dz)
                key = (arr, cost)
                if bestk is None or key < bestk:
                    bestk, tree = key, tr
            choice_cache[ck] = tree
        new_id[nid] = build(tree, leaves)
    out.input_node = {s: new_id[v] for s, v in net.input_node.items()}
    out.sig_node = np.asarray([new_id[int(v)] for v in net.sig_node], dtype=np.int64)
    out.asserts = [new_id[a] for a in net.asserts]
    n_prims = sum(1 for k in kind if k >= 0)
    out.stats = dict(net.stats)
    out.stats.update({"lut_gates": net.stats.get("gates"), "lut_depth": net.stats.get("depth"), "prims": n_prims,
                      "depth": max(level) if level else 0})
    return out


def simulate(net: PrimNet, input_masks: dict, width: int):
    """reference evaluation of a PrimNet on `width` instances (python ints as masks); input_masks: signal -> mask"""
    full = (1 << width) - 1
    val = [0] * len(net.kind)
    val[1] = full
    for s, nid in net.input_node.items():
        val[nid] = input_masks[s] & full
    for nid in range(2, len(net.kind)):
        k = net.kind[nid]
        if k < 0:
            continue
        p, q, r = val[net.a[nid]], val[net.b[nid]], val[net.c[nid]]
        u = (p & q) if (k & K_AND) else (p ^ q)
        val[nid] = (u | r) if (k & K_OR) else (u ^ r)
    return val
