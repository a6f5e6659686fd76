"""Pass P: the latency-hiding single-wave schedule ("pipelined variant", csrc/cw_kernels.hip cw_pipe_kernel).

Why.  In the strand schedules (lower.py passes C/D) every operand that is not the previous row's result is read from
the value table in HBM/L2 at most one row ahead of its use.  With one or two waves per SIMD (small batches of deep
circuits: EdDSA / Merkle at batch 8 192, Poseidon at batch 65 536) nothing covers that latency: measured ~3 000 clocks
per row against ~330 for the Montgomery product itself (profiles/, tools/profile_ops.sh).  Here the rows of ONE wave are
organised so that no row ever waits for the value table:

  * rows come in batches of NB.  The far operands of batch k (value-table slots and constants) are named in the batch's
    LOAD LIST (at most NLD entries) and copied global -> LDS by the LDS-DMA path one whole batch ahead: L(k) is issued
    when batch k-1 starts and awaited when batch k starts.  They land in staging entries (double buffered: half k % 2);
  * every result is also written to a RING of RR = 2*NB LDS entries (entry = row position mod RR), so a consumer at most
    RR rows behind its producer reads LDS; the previous row's result is forwarded in registers (PREV) as before;
  * a consumer more than RR rows behind reads the value table through the load list of its batch: the producer's store
    (row p) precedes the issue of L(batch(c)) (start of row (batch(c)-1)*NB) because c - p > 2*NB;
  * every row carries exactly two store targets (value-table slots or NONE = dropped by the buffer bounds check) and a
    batch exactly NLD loads, so the number of vector-memory instructions between the issue of L(k) and the start of
    batch k is a constant (4*NB) and the kernel's only wait is `s_waitcnt vmcnt(4*NB)`.

The planner walks the single-strand row stream in program order, classifies every operand PREV / RING / STAGED, closes a
batch early (NOP rows) when its load list is full, splits long LINSUM / DOTC rows into chains and adds pure store rows
for values with more than two destinations.  oracle/tape_eval.py eval_pipe replays the result with the kernel's timing.
"""
from __future__ import annotations

import numpy as np

from .. import opcodes as O

K_SIG, K_TMP, K_CONST, K_NONE = O.K_SIG, O.K_TMP, O.K_CONST, O.K_NONE
KO_PREV, K_LDS = 3, 4
P_NONE = 0xFFFFFFFF
PX_TMP, PX_CONST = 1 << 31, 1 << 30
ENTRY_NONE = 0xFF
D_NOP = 255
FORCE_PREV = -1                     # operand kind inside the planner only (K_NONE and the device's PREV share the number 3)

DEFAULT_NB, DEFAULT_NLD = 8, 8


class PipeError(Exception):
    pass


def plan_pipe(stream, n_signals, dconsts, D, nb=DEFAULT_NB, nld=DEFAULT_NLD):
    """stream: list of lower._Row in program order (single strand, after aliasing).  D: the lower module (opcodes, _Row).
    Returns dict(rows [n,8] u32, loads [n_batches+2, nld] u32, terms list of (kind, entry, signed coef | lconst key),
    n_tslots, stats)."""
    if nb not in (4, 8) or nld not in (4, 8):
        raise PipeError("NB and NLD must be 4 or 8")
    RR = 2 * nb
    tmax = nld - 1                      # terms per LINSUM / DOTC row (all of them may be far, plus the constant term)
    NOVAL = D._NO_VALUE

    def vid(k, v):
        return v if k == K_SIG else n_signals + v

    # ---- names with a non-forwarded use (superset of the temps that must reach the value table) -----------------------
    # (exact PREV-ness depends on the final positions; a temp stored needlessly is pruned after placement)
    used = set()
    for r in stream:
        for k, v in D._value_operands(r):
            if k == K_TMP:
                used.add(v)

    out = []            # placed rows: dict(op, a, b, flag, aux, names, stores, terms, cmag)
    batch_loads = []    # per batch: list of sources ('v', value id) | ('c', const index)
    prod = {}           # value id -> position of the row that (last) holds it in its ring entry
    last_names = ()     # names of the last value-producing row (PREV)
    far_names = set()   # value ids that are loaded from the value table somewhere
    n_pad = 0

    def cur_batch():
        k = len(out) // nb
        while len(batch_loads) <= k:
            batch_loads.append([])
        return k

    def close_batch():
        nonlocal n_pad
        while len(out) % nb:
            out.append(None)
            n_pad += 1

    def classify(ops, pos, k, newloads):
        """ops: list of (kind, id).  Returns list of (KO_PREV|K_LDS|0, entry) or None when the load list overflows."""
        res = []
        have = batch_loads[k]
        for kk, vv in ops:
            if kk == FORCE_PREV:        # forced forwarding (running sum of a split LINSUM / DOTC)
                res.append((KO_PREV, 0))
                continue
            if kk == K_NONE:
                res.append((0, 0))
                continue
            if kk == K_CONST:
                src = ('c', vv)
            else:
                x = vid(kk, vv)
                if x in last_names:
                    res.append((KO_PREV, 0))
                    continue
                p = prod.get(x)
                if p is not None and pos - p <= RR:
                    res.append((K_LDS, p % RR))
                    continue
                src = ('v', x)
            if src in have:
                j = have.index(src)
            elif src in newloads:
                j = len(have) + newloads.index(src)
            else:
                if len(have) + len(newloads) >= nld:
                    return None
                newloads.append(src)
                j = len(have) + len(newloads) - 1
            res.append((K_LDS, RR + (k & 1) * nld + j))
        return res

    def place(op, ops, terms, names, stores, flag=0, aux=0, cmag=0, seq=0):
        """terms: list of (kind, id, coef) or None.  names: value ids this row's result is known by."""
        nonlocal last_names
        while True:
            pos = len(out)
            k = cur_batch()
            newloads = []
            cl = classify(ops, pos, k, newloads)
            tl = classify([(t[0], t[1]) for t in terms], pos, k, newloads) if (cl is not None and terms) else []
            if terms is not None and len(terms) > tmax:
                raise PipeError("too many terms in one row")
            if cl is None or tl is None:
                if pos % nb == 0:
                    raise PipeError("a row needs more than %d loads" % nld)
                close_batch()
                continue
            break
        for s in newloads:
            if s[0] == 'v':
                far_names.add(s[1])
        batch_loads[k].extend(newloads)
        row = dict(op=op, a=cl[0] if len(cl) > 0 else (0, 0), b=cl[1] if len(cl) > 1 else (0, 0), flag=flag, aux=aux, seq=seq,
                   names=tuple(names), stores=list(stores[:2]), cmag=cmag,
                   terms=[(c_[0], c_[1], t[2]) for c_, t in zip(tl, terms)] if terms is not None else None)
        out.append(row)
        if op not in NOVAL and op != D_NOP:
            for x in names:
                prod[x] = pos
            last_names = frozenset(names)
        rest = stores[2:]
        while rest:                     # further destinations: pure store rows (d = PREV)
            pos = len(out)
            cur_batch()
            out.append(dict(op=D.D_COPY, a=(KO_PREV, 0), b=(0, 0), flag=0, aux=0, names=tuple(names), stores=list(rest[:2]),
                            cmag=0, terms=None))
            for x in names:
                prod[x] = pos
            rest = rest[2:]

    for r in stream:
        if r == "B":
            raise PipeError("barrier in a single-strand stream (function call)")
        op = r.op
        if op == D.D_CALL:
            raise PipeError("circom functions with run-time control flow are not pipelined")
        names, stores = [], []
        if op not in NOVAL:
            dsts = []
            if r.dk in (K_SIG, K_TMP):
                dsts.append((r.dk, r.dv))
            if r.extra:
                dsts.extend(r.extra)
            for kk, vv in dsts:
                names.append(vid(kk, vv))
                if kk == K_SIG or vv in used:
                    stores.append(vid(kk, vv))
        if op in (D.D_LINSUM, D.D_DOTC):
            terms = [tuple(t) for t in r.terms]
            c0 = (r.bk, r.bv) if r.bk == K_CONST else (K_NONE, 0)
            chunks = [terms[i:i + tmax] for i in range(0, len(terms), tmax)] or [[]]
            for ci, ch in enumerate(chunks):
                last = ci == len(chunks) - 1
                # later chunks add the running sum, forwarded in registers: operand b = PREV
                b = c0 if ci == 0 else (FORCE_PREV, 0)
                place(op, [(K_NONE, 0), b], ch, names if last else [], stores if last else [], aux=len(ch))
            continue
        if op == D.D_BIT:
            place(op, [(r.ak, r.av)], None, names, stores, aux=r.bv)
            continue
        if op in (D.D_MULC, D.D_MADDC):
            cmag = dconsts[r.bv + 1] if r.flag else 0
            place(op, [(r.ak, r.av), (r.bk, r.bv)], None, names, stores, flag=r.flag, cmag=cmag)
            continue
        place(op, [(r.ak, r.av), (r.bk, r.bv)], None, names, stores, seq=r.seq)
    close_batch()
    n_rows = len(out)
    n_batches = n_rows // nb
    while len(batch_loads) < n_batches + 2:
        batch_loads.append([])

    # ---- value-table slots of the temps that are really loaded; liveness = producer position .. issue of the last load --
    last_issue = {}
    for k, lst in enumerate(batch_loads):
        for s in lst:
            if s[0] == 'v' and s[1] >= n_signals:
                last_issue[s[1]] = max(0, (k - 1) * nb)       # L(k) reads the table when batch k-1 starts
    slot_of = {}
    free, release = [], {}
    n_tslots = 0
    for pos, row in enumerate(out):
        for x in release.pop(pos, ()):
            free.append(slot_of[x])
        if row is None:
            continue
        for x in row["stores"]:
            if x >= n_signals and x in last_issue and x not in slot_of:
                if free:
                    sl = free.pop()
                else:
                    sl = n_tslots
                    n_tslots += 1
                slot_of[x] = sl
                # reusable by a row whose store is issued after the last load of x was: positions > last_issue
                release.setdefault(max(last_issue[x], pos) + 1, []).append(x)

    # ---- encode --------------------------------------------------------------------------------------------------------
    rows = np.zeros((n_rows, 8), dtype=np.uint32)
    terms_out = []
    n_prev = n_ring = n_staged = n_stores = 0
    for pos, row in enumerate(out):
        if row is None:
            rows[pos] = (D_NOP, 0, ENTRY_NONE << 16, P_NONE, P_NONE, 0, 0, 0)
            continue
        (ak, ae), (bk, be) = row["a"], row["b"]
        op = row["op"]
        value = op not in NOVAL
        st = []
        for x in row["stores"]:
            if x < n_signals:
                st.append(x)
            elif x in slot_of:
                st.append(PX_TMP | slot_of[x])
        st += [P_NONE] * (2 - len(st))
        n_stores += sum(1 for s in st if s != P_NONE)
        aux = row["aux"]
        if op in (D.D_ASSERT_EQ, D.D_ASSERT_NZ, D.D_IDIV, D.D_MOD):
            aux = row.get("seq", 0)         # index of the flat operation: what the status word reports
        for kk, ee in ((ak, ae), (bk, be)):
            n_prev += kk == KO_PREV
            n_ring += kk == K_LDS and ee < RR
            n_staged += kk == K_LDS and ee >= RR
        if row["terms"] is not None:
            for (tk, te, cf) in row["terms"]:
                terms_out.append((tk, te, cf, op == D.D_DOTC))
                n_prev += tk == KO_PREV
                n_ring += tk == K_LDS and te < RR
                n_staged += tk == K_LDS and te >= RR
        d_entry = (pos % RR) if value else ENTRY_NONE
        cm = row["cmag"]
        rows[pos] = (op | (ak << 8) | (bk << 11) | (row["flag"] << 29), aux, ae | (be << 8) | (d_entry << 16), st[0], st[1],
                     cm & 0xFFFFFFFF, cm >> 32, 0)
    loads = np.full((n_batches + 2, nld), P_NONE, dtype=np.uint32)
    n_loads = 0
    for k, lst in enumerate(batch_loads[:n_batches + 2]):
        for j, s in enumerate(lst):
            if s[0] == 'c':
                loads[k, j] = PX_CONST | s[1]
            elif s[1] < n_signals:
                loads[k, j] = s[1]
            else:
                loads[k, j] = PX_TMP | slot_of[s[1]]
            n_loads += 1
    stats = {"pipe_rows": n_rows, "pipe_pad": n_pad, "pipe_loads": n_loads, "pipe_prev": int(n_prev), "pipe_ring": int(n_ring),
             "pipe_staged": int(n_staged), "pipe_stores": n_stores, "pipe_nb": nb, "pipe_nld": nld}
    return dict(rows=rows, loads=loads, terms=terms_out, n_tslots=n_tslots, stats=stats, ring=RR)
