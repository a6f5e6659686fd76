"""`--O1`: the constraint simplification the reference applies BY DEFAULT (circom/src/input_user.rs:264-283: no flag = O1) -
"only constant and renaming (equalities between signals) simplifications" (mkdocs formats/constraints-json.md) - restated
from constraint_list/src/constraint_simplification.rs `simplification` with flag_s = true (apply_linear = false):

  1. the constraints of the flat circuit are sorted into constant equalities (k s + m = 0), signal equalities (k (s0 - s1) = 0),
     other linear ones and non-linear ones (dag/src/map_to_constraint_list.rs:27-36; circom_algebra algebra.rs:1346-1372);
  2. signal equalities form clusters (connected components); a cluster is replaced by ONE representative - its smallest
     FORBIDDEN signal (wire 0, main's outputs, main's PUBLIC inputs: dag/src/lib.rs:174-198) or, if it has none, its smallest
     signal; its other forbidden signals keep a constraint `s - representative = 0`, every other signal is substituted
     (eq_cluster_simplification :126-196; a cluster of one constraint between two forbidden signals keeps that constraint as it is);
  3. constant equalities of non-forbidden signals become substitutions signal -> constant (constant_eq_simplification :253-273);
  4. both substitution maps are applied to the linear and the non-linear constraints; a product whose factor became a constant
     folds into a linear constraint (fix_raw_constraint, algebra.rs:1309-1344);
  5. the new system is: non-linear constraints in tree order, those that became linear, the kept equalities, the kept constant
     equalities, the linear constraints; empty ones are dropped (:588-693);
  6. the witness keeps, in their old order, the signals that are neither substituted nor unused (in no constraint and not
     forbidden): rebuild_witness :101-124 is a stable compaction.  Private inputs CAN disappear (the header's nPrvIn shrinks).

What the device does with it: nothing changes on the device - it generates and checks the FULL (--O0) system, which implies the
simplified one; the O1 witness is the O0 witness read through `witness2signal` (reduce_wtns below, or a `.dat` written with that
list: the reference runtime then writes exactly those bytes - tests/test_circom_simplify.py runs it).  The `.r1cs` / `.sym` /
constraints JSON a prover takes are written from the simplified system.  `--O2` (Gaussian elimination of linear constraints
with the reference's signal-choice heuristics) is not implemented.
"""
from __future__ import annotations

import struct

import numpy as np


class Simplified:
    """constraints over the NEW wire numbering, witness2signal (kept signals in order), signal2witness (-1 = removed)"""

    def __init__(self, fc, constraints, w2s, n_prv_in, substituted, constants, unused):
        self.fc = fc
        self.constraints = constraints
        self.witness2signal = w2s
        self.signal2witness = np.full(fc.n_signals, -1, dtype=np.int64)
        self.signal2witness[np.asarray(w2s, dtype=np.int64)] = np.arange(len(w2s))
        self.n_wires = len(w2s)
        self.n_prv_in = n_prv_in
        self.substituted = substituted        # removed signal -> representative signal
        self.constants = constants            # removed signal -> constant value
        self.unused = unused                  # removed because no constraint mentions them


def _subst(d, eq, cn, q):
    """apply signal -> signal and signal -> constant substitutions to one linear form (keys: signals, 0 = the constant)"""
    hit = False
    for k in d:
        if k in eq or k in cn:
            hit = True
            break
    if not hit:
        return d
    out = {}
    for k, v in d.items():
        if k in eq:
            k2 = eq[k]
            if k2 in cn:                                       # (a representative that a constant equality removed)
                out[0] = (out.get(0, 0) + v * cn[k2]) % q
            else:
                out[k2] = (out.get(k2, 0) + v) % q
        elif k in cn:
            out[0] = (out.get(0, 0) + v * cn[k]) % q
        else:
            out[k] = (out.get(k, 0) + v) % q
    return out


def _fix(a, b, c, q):
    """fix_raw_constraint (algebra.rs:1309-1344)"""
    a = {k: v for k, v in a.items() if v % q}
    b = {k: v for k, v in b.items() if v % q}
    c = {k: v for k, v in c.items() if v % q}
    if not a or not b:
        return {}, {}, c
    for x, y in ((a, b), (b, a)):
        if len(x) == 1 and 0 in x:                             # a constant factor: k * y - c = 0  ->  c - k y
            k = x[0]
            c = dict(c)
            for s, v in y.items():
                c[s] = (c.get(s, 0) - k * v) % q
            return {}, {}, {s: v for s, v in c.items() if v}
    return a, b, c


def simplify_o1(fc) -> Simplified:
    q = fc.fp.q
    n = fc.n_signals
    forbidden = {0} | set(range(1, 1 + fc.n_outputs)) | set(range(fc.main_input_start, fc.main_input_start + fc.n_pub_in))
    eqs, cons_eqs, linear, nonlinear = [], [], [], []
    for a, b, c in fc.constraints:
        if not a and not b:
            if (0 in c and len(c) == 2) or (0 not in c and len(c) == 1):
                cons_eqs.append(dict(c))
                continue
            if 0 not in c and len(c) == 2:
                (s0, v0), (s1, v1) = c.items()
                if (v0 + v1) % q == 0:
                    eqs.append(dict(c))
                    continue
            linear.append(dict(c))
        else:
            nonlinear.append((dict(a), dict(b), dict(c)))

    # ---- clusters of signal equalities (build_clusters :45-99: a cluster ends at the arena slot of its LAST constraint) ----------
    parent = {}

    def find(x):
        r = x
        while parent[r] != r:
            r = parent[r]
        while parent[x] != r:
            parent[x], x = r, parent[x]
        return r
    for c in eqs:
        s0, s1 = c.keys()
        for s in (s0, s1):
            parent.setdefault(s, s)
        r0, r1 = find(s0), find(s1)
        if r0 != r1:
            parent[r0] = r1
    members, last, count = {}, {}, {}
    for i, c in enumerate(eqs):
        r = find(next(iter(c)))
        members.setdefault(r, set()).update(c.keys())
        last[r] = i
        count[r] = count.get(r, 0) + 1
    eq_sub, kept_eq = {}, []
    for r in sorted(members, key=lambda r: last[r]):
        sigs = sorted(members[r])
        if count[r] == 1:
            s0, s1 = sigs
            if s0 in forbidden and s1 in forbidden:
                kept_eq.append(eqs[last[r]])
            elif s0 in forbidden:
                eq_sub[s1] = s0
            elif s1 in forbidden:
                eq_sub[s0] = s1
            else:
                eq_sub[s1] = s0                                  # the larger signal goes
            continue
        remains = [s for s in sigs if s in forbidden]
        rh = remains[0] if remains else sigs[0]
        for s in remains:
            if s != rh:
                kept_eq.append({s: q - 1, rh: 1})               # transform_expression_to_constraint_form(s - rh): C = -(s - rh)
        for s in sigs:
            if s not in forbidden and s != rh:
                eq_sub[s] = rh

    # ---- constant equalities ------------------------------------------------------------------------------------------------
    none = {}
    linear = [_subst(c, eq_sub, none, q) for c in linear]
    cons_eqs = [{k: v for k, v in _subst(c, eq_sub, none, q).items() if v} for c in cons_eqs]
    cn_sub, kept_cons = {}, []
    for c in cons_eqs:
        sigs = sorted(k for k in c if k != 0)
        if not sigs:
            kept_cons.append(c)                                  # (both sides constant after a substitution: kept as it is)
            continue
        s = sigs[-1]
        if s in forbidden:
            kept_cons.append(c)
        else:
            cn_sub[s] = (-c.get(0, 0) * pow(c[s], q - 2, q)) % q  # k s + m = 0
    linear = [_subst(c, none, cn_sub, q) for c in linear]

    # ---- the new system ---------------------------------------------------------------------------------------------------------
    store, became_linear = [], []
    for a, b, c in nonlinear:
        a, b, c = (_subst(x, eq_sub, cn_sub, q) for x in (a, b, c))
        a, b, c = _fix(a, b, c, q)
        (store if a else became_linear).append((a, b, c))
    store += became_linear
    for c in kept_eq + kept_cons + linear:
        store.append(_fix({}, {}, c, q))
    store = [(a, b, c) for a, b, c in store if a or b or c]

    used = set()
    for a, b, c in store:
        used.update(a)
        used.update(b)
        used.update(c)
    deleted = set(eq_sub) | set(cn_sub)
    w2s, unused = [], []
    for s in range(n):
        if s in deleted:
            continue
        if s not in forbidden and s not in used:
            unused.append(s)
            continue
        w2s.append(s)
    pos = {s: i for i, s in enumerate(w2s)}
    cons = [({pos[k]: v for k, v in a.items()}, {pos[k]: v for k, v in b.items()}, {pos[k]: v for k, v in c.items()})
            for a, b, c in store]
    lo, hi = fc.main_input_start, fc.main_input_start + fc.n_pub_in + fc.n_prv_in
    gone_inputs = sum(1 for s in list(deleted) + unused if lo <= s < hi)
    return Simplified(fc, cons, w2s, fc.n_prv_in - gone_inputs, eq_sub, cn_sub, unused)


# ---- writers (same layouts as hip_elements/writers.py; the O0 writers stay untouched) ---------------------------------------------
def write_r1cs(path, sm: Simplified):
    """constraint_list/src/r1cs_porting.rs: header with the NEW wire count and the shrunken nPrvIn, nLabels = all signals,
    wire2label = witness2signal"""
    from ..hip_elements.writers import _lc_block
    fc = sm.fc
    q = fc.fp.q
    bits = q.bit_length()
    fs = bits // 8 if bits % 64 == 0 else (bits // 64 + 1) * 8
    sec2 = b"".join(_lc_block(a, fs) + _lc_block(b, fs) + _lc_block(c, fs) for a, b, c in sm.constraints)
    sec1 = struct.pack("<I", fs) + q.to_bytes(fs, "little") + struct.pack(
        "<IIIIQI", sm.n_wires, fc.n_outputs, fc.n_pub_in, sm.n_prv_in, fc.n_signals, len(sm.constraints))
    sec3 = np.asarray(sm.witness2signal, dtype="<u8").tobytes()
    with open(path, "wb") as f:
        f.write(b"r1cs" + struct.pack("<II", 1, 3))
        for typ, body in ((2, sec2), (1, sec1), (3, sec3)):
            f.write(struct.pack("<IQ", typ, len(body)))
            f.write(body)


def write_sym(path, sm: Simplified):
    """constraint_list/src/sym_porting.rs: <label>,<witness position or -1>,<component>,<name>"""
    import io
    import os
    from ..hip_elements.writers import write_sym as write_sym_o0
    tmp = str(path) + ".o0"
    write_sym_o0(tmp, sm.fc)
    s2w = sm.signal2witness
    out = io.StringIO()
    with open(tmp) as f:
        for line in f:
            s, _w, rest = line.split(",", 2)
            out.write("%s,%d,%s" % (s, s2w[int(s)], rest))
    os.remove(tmp)
    with open(path, "w") as f:
        f.write(out.getvalue())


def constraints_json(constraints) -> str:
    """constraint_writers/src/json_writer.rs: {"constraints": [[A, B, C], ...]} with decimal strings, wires ascending"""
    rows = []
    for con in constraints:
        rows.append("[" + ",".join("{" + ",".join('"%d":"%d"' % (w, v) for w, v in sorted(part.items())) + "}" for part in con) + "]")
    return '{\n"constraints": [\n' + ",\n".join(rows) + "\n]\n}"


def reduce_wtns(data: bytes, w2s) -> bytes:
    """the `.wtns` of the simplified system from the `.wtns` of the full one: section 1 with the new count, section 2 = the kept
    entries in order (main.cpp:288-334 writes exactly this when its `.dat` carries the list)"""
    assert data[:4] == b"wtns"
    n8 = struct.unpack_from("<I", data, 24)[0]
    hdr1 = 12 + 12 + 4 + n8                    # magic, version, nSections | id, len | n8 | q
    n_old = struct.unpack_from("<I", data, hdr1)[0]
    body0 = hdr1 + 4 + 12
    assert len(data) == body0 + n_old * n8
    vals = np.frombuffer(data, dtype=np.uint8, count=n_old * n8, offset=body0).reshape(n_old, n8)
    sel = vals[np.asarray(w2s, dtype=np.int64)]
    return (data[:hdr1] + struct.pack("<I", len(w2s)) + struct.pack("<IQ", 2, len(w2s) * n8) + sel.tobytes())
