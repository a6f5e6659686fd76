"""hip_elements bit-plane lowering, part 2: primitive network (bitmap.PrimNet) -> the program of `cw_bits_eval_kernel`.

Execution model (csrc/cw_bits.hip), round 3.  One wave evaluates the whole network for ONE group of 64 instances; a
value is a mask (bit i = the value in instance i of the group).  The program is a sequence of VROWS; in a vrow every LANE
evaluates one two-stage primitive (bitmap.py) on three masks it reads from the wave's LDS and writes the result to ONE
LDS entry.  The wave's LDS holds

    ring    R rows x 64 entries   results that are no signal: entry (vrow mod R, lane), overwritten R vrows later
    cache   C rows x 64 entries   ROWS of the group's bit table `T[group][row * 64 + pos]` (HBM): rows being filled by
                                  the program (every signal value has a home entry = its bit-table slot), rows kept after
                                  they were flushed, rows loaded back from the table (main inputs, old values)
    consts  2 entries             0 and ~0

so every operand is an LDS entry and vector memory is touched only by whole-row traffic at batch boundaries (a batch = 8
vrows): 512-byte coalesced row LOADS and row FLUSHES named in a per-batch command block (scalar loads), plus the record
stream itself (8 bytes per lane and vrow).  Round 2 stored every result through a per-lane scattered 8-byte store and
fetched old values through per-lane scattered loads: 2.5 vector-memory instructions per vrow and wave, measured at
~23-48 clocks of the CU's memory pipeline each with four waves per CU (tools/ubench_isa) = the bound of that kernel.

Timing contract (mirrored by oracle/tape_eval.py::eval_bits and checked there):
  * operands of vrow v+1 are read BEFORE vrow v writes its result: a consumer sits >= LATENCY = 2 vrows behind its producer;
  * a ring entry lives R - 1 vrows;
  * batch b: row loads of cmd[b] are requested when the batch starts and written to their cache slots between steps 6
    and 7 of batch b + 1 (readable by every vrow of batch b + 2); flushes of cmd[b] copy a cache slot to the table DURING
    batch b + 1 (the kernel reads the first two slots before that batch's first vrow and the others after it: nothing may
    overwrite a flushed slot before batch b + 2); a load of batch b sees flushes of batches <= b - 2.
A row is flushed exactly once, when all its positions are produced (never read-modify-write).  Signals that the R1CS
check reads as whole 32-bit words (32 consecutive signals) keep 32 consecutive slots: the first of them to be produced
reserves half a row for all of them (ATOMS); the row stays pinned until they all arrived.

Signals that are copies of one another are the same value: they share a slot (`sig_slot[]`).
Bit-table slots: 0 = constant 0, 1 = constant 1 (all ones), 2 = reserved, main input k = IN_BASE + k (rows filled by the
init / ingest kernels), then the rows the program produces.

Record (2 x u32 per lane):  a_off | K1 | K2 << 1 | b_off << 16,   c_off | dst_off << 16      (LDS byte offsets, multiples of 8)
Command block (CMD_WORDS u32 per batch):  n_loads | n_flushes << 8, 0, MAX_LOADS x (table byte offset of the row, LDS
byte offset of the slot), MAX_FLUSH x (same), padding.
"""
from __future__ import annotations

import heapq

import numpy as np

from .bitmap import PrimNet, K_AND, K_OR

IN_BASE = 3                   # slot of main input 0
DEFAULT_RING = 32             # R
DEFAULT_CACHE = 44            # C: (R + C) * 512 + 16 bytes of LDS per wave; four waves per CU fit 160 KiB
LATENCY = 2
BATCH = 8                     # vrows per batch (cw_bits.hip BITS_NB)
MAX_LOADS = 4                 # row loads per batch (two register sets of that many masks in the kernel)
MAX_FLUSH = 6                 # row flushes per batch
CMD_WORDS = 24
LOAD_DELAY = 2                # a row requested in batch b is readable from batch b + LOAD_DELAY
ATOM = 32                     # signals the R1CS check reads as one word
ATOM_SPREAD = 64              # levels: an atom whose members are further apart than this is not kept together
PREFETCH_PENDING = -1         # rows of an operation's operands are requested when this many of its producers are still missing (-1: on demand only)
WINDOW = 8                    # levels an operation may run ahead of the frontier
CLOSE_AGE = 64                # vrows after which a half-filled atom row with nothing pending is closed


import os as _os
_DEBUG = bool(_os.environ.get("CW_SCHED_DEBUG"))


class BitTape:
    def __init__(self):
        self.ring = DEFAULT_RING        # R: ring rows
        self.cache = DEFAULT_CACHE      # C: cache slots (rows)
        self.n_slots = 0                # bit-table slots per group (multiple of 64)
        self.n_vrows = 0                # multiple of BATCH
        self.recs = None                # uint32 [n_vrows * 64, 2]
        self.cmds = None                # uint32 [n_vrows / BATCH, CMD_WORDS]
        self.sig_slot = None            # uint32 [n_signals]: slot holding each signal
        self.assert_slots = None        # uint32 [n_asserts]: slots that must be 0 in every instance
        self.n_signals = 0
        self.n_inputs = 0
        self.input_start = 0
        self.stats = {}
        self.saved = set()              # (scheduler-internal) values that were copied to a home by a SAVE lane


def lower_bits(net: PrimNet, fc, ring: int = DEFAULT_RING, cache: int = DEFAULT_CACHE, passes: int = 2, window: int = None):
    """Returns the BitTape, or None when the network has nothing to evaluate (every signal an input or a constant).
    A value that is no signal lives in the ring; when it is about to leave the ring with consumers still waiting, a SAVE
    lane copies it to a bit-table home (one extra lane).  Pass 1 finds those values; pass 2 gives them their home from
    the start (no extra lane) and saves the few that the changed schedule adds."""
    homes = set()
    bt = None
    for _ in range(max(1, passes)):
        bt = _schedule(net, fc, ring, cache, homes, WINDOW if window is None else window)
        if bt is None or not bt.saved:
            break
        homes |= bt.saved
    return bt


def _schedule(net: PrimNet, fc, ring: int, cache: int, extra_homes: set, WINDOW: int):
    assert 8 <= ring <= 96 and ring % BATCH == 0 and cache >= 8 and (ring + cache) * 512 + 16 <= 0xFFF8
    kind, A, B, C = net.kind, net.a, net.b, net.c
    n_nodes = len(kind)
    n_signals = fc.n_signals
    sig_node = net.sig_node
    n_in = fc.n_main_inputs
    is_gate = [k >= 0 for k in kind]
    is_signal = np.zeros(n_nodes, dtype=bool)
    is_signal[sig_node] = True
    live = is_signal.tolist()
    for a in net.asserts:
        live[a] = True
    for nid in range(n_nodes - 1, 1, -1):
        if live[nid] and is_gate[nid]:
            live[A[nid]] = live[B[nid]] = live[C[nid]] = True
    is_gate = [g and live[i] for i, g in enumerate(is_gate)]
    n_gates = sum(is_gate)
    if n_gates == 0:
        return None
    stored = [bool(is_gate[i] and (is_signal[i] or i in extra_homes)) for i in range(n_nodes)]
    for a in net.asserts:
        if is_gate[a]:
            stored[a] = True

    # ---- homes of constants and inputs; atoms -----------------------------------------------------------------------------
    home = {0: 0, 1: 1}                       # node -> bit-table slot
    for s, nid in net.input_node.items():
        home[nid] = IN_BASE + (s - fc.main_input_start)
    n_in_rows = (IN_BASE + n_in + 63) // 64
    first_sig = {}
    for s_ in range(n_signals - 1, -1, -1):
        first_sig[int(sig_node[s_])] = s_
    level = net.level
    atom_of = {}                              # node -> (atom id, index)
    atoms = []                                # [base slot or None, members]
    # (1) words the R1CS check reads whole (cw_bits_host.h: 32 consecutive slots carrying 2^0 .. 2^31 of one sign inside one
    #     32-bit half of a long linear row - BinSum's `lin === lout`, Bits2Num): taken from the constraints themselves
    q = fc.fp.q
    pow2 = {}
    for k in range(64):
        pow2[1 << k] = (0, k)
        pow2[q - (1 << k)] = (1, k)
    for cons in getattr(fc, "constraints", ()):
        if len(cons[0]) + len(cons[1]) + len(cons[2]) < 33:
            continue
        for part in cons:
            if len(part) < 32:
                continue
            words = {}
            for sgn, cf in part.items():
                pk = pow2.get(cf)
                if pk is not None:
                    words.setdefault((pk[0], pk[1] // 32), {})[pk[1] % 32] = sgn
            for w in words.values():
                if len(w) != 32:
                    continue
                mem = [int(sig_node[w[k]]) for k in range(32)]
                if len(set(mem)) != 32 or any((not stored[n]) or n in atom_of for n in mem):
                    continue
                lv = [level[n] for n in mem]
                if max(lv) - min(lv) > ATOM_SPREAD:
                    continue
                for j, n in enumerate(mem):
                    atom_of[n] = (len(atoms), j)
                atoms.append([None, mem])
    # (2) runs of 32 consecutive signals among the rest (the check's blocks of 8 single-bit terms want consecutive slots too)
    sg = sorted((first_sig[i], i) for i in range(2, n_nodes) if stored[i] and i in first_sig and i not in atom_of)
    run = []

    def close_run():
        for k in range(0, len(run) - ATOM + 1, ATOM):
            mem = [n for _, n in run[k:k + ATOM]]
            lv = [level[n] for n in mem]
            if max(lv) - min(lv) <= ATOM_SPREAD:
                for j, n in enumerate(mem):
                    atom_of[n] = (len(atoms), j)
                atoms.append([None, mem])

    for fs, nid in sg:
        if run and fs != run[-1][0] + 1:
            close_run()
            run = []
        run.append((fs, nid))
    close_run()

    # ---- operations ----------------------------------------------------------------------------------------------------------
    height = [0] * n_nodes
    for nid in range(n_nodes - 1, 1, -1):
        if not is_gate[nid]:
            continue
        h = height[nid] + 1
        for o in (A[nid], B[nid], C[nid]):
            if is_gate[o] and height[o] < h:
                height[o] = h
    ops = [nid for nid in range(2, n_nodes) if is_gate[nid]]
    consumers = {}
    pending = {}
    for nid in ops:
        srcs = {x for x in (A[nid], B[nid], C[nid]) if is_gate[x]}
        pending[nid] = len(srcs)
        for x in srcs:
            consumers.setdefault(x, []).append(nid)
    # target level of every operation: as late as its earliest consumer allows (an operation nobody reads - most signals -
    # runs as soon as it can, which frees its operands); an operation is held back while its target is more than
    # WINDOW levels ahead of the frontier (= the highest target placed so far), so that values are produced near their
    # consumers: SHA-256's message schedule would otherwise be computed hundreds of vrows before the rounds that read
    # it, pass through the ring into bit-table rows and come back through row loads.
    # The anchor is the cone of the circuit's outputs (and assertions): inside it an operation's target is set by its
    # earliest consumer of the cone; everything outside - signals nobody reads, the gates only they need - follows its
    # producers (target = theirs + 1: the operands are at hand then).  Anchoring such dead-end signals at their ASAP level
    # instead would drag their producers, e.g. the message schedule of EVERY block, to the front of the program.
    in_cone = [False] * n_nodes
    for s_ in range(1, fc.main_input_start):
        in_cone[int(sig_node[s_])] = True
    for a in net.asserts:
        in_cone[a] = True
    for nid in reversed(ops):
        if in_cone[nid]:
            in_cone[A[nid]] = in_cone[B[nid]] = in_cone[C[nid]] = True
    target = {}
    for nid in reversed(ops):
        if in_cone[nid]:
            cs = [target[c] for c in consumers.get(nid, ()) if in_cone[c]]
            target[nid] = max(level[nid], min(cs) - 1) if cs else level[nid]
    for nid in ops:
        if not in_cone[nid]:
            ps = [target[p] for p in (A[nid], B[nid], C[nid]) if is_gate[p]]
            target[nid] = max(level[nid], max(ps) + 1) if ps else level[nid]
    heap = [(target[nid], -height[nid], nid) for nid in ops if pending[nid] == 0]
    heapq.heapify(heap)
    buckets = {}
    frontier = 0
    remaining = {x: len(v) for x, v in consumers.items()}      # consumers not placed yet
    expiry = {}                                                # vrow -> ring values to look at (SAVE if still needed)
    input_refs = {}
    for nid in ops:
        for x in {A[nid], B[nid], C[nid]}:
            if x > 1 and not is_gate[x]:
                input_refs[x] = input_refs.get(x, 0) + 1
    SAVE_AT = ring - 1 - LATENCY - 2                           # vrows after production
    assert SAVE_AT >= LATENCY
    saved = set()

    # ---- LDS state --------------------------------------------------------------------------------------------------------------
    ring_bytes = ring * 512
    const_off = (ring + cache) * 512
    FREE, OPEN, CLEAN, LOADING = 0, 1, 2, 3
    slot_state = [FREE] * cache
    slot_row = [-1] * cache
    slot_last = [-1] * cache                  # last vrow that read or wrote the slot
    slot_ready = [0] * cache                  # first vrow that may read a LOADING / CLEAN-after-load slot
    slot_pend = [0] * cache                   # reserved positions not produced yet (> 0 = pinned)
    slot_used = [0] * cache                   # positions handed out
    slot_open_at = [0] * cache
    slot_dirty = [False] * cache
    slot_free_at = [0] * cache                # first vrow that may overwrite the slot (a flushed row is read by the batch AFTER its flush)
    resident = {}                             # row -> cache slot
    row_flushed = {}                          # row -> batch of its flush (input rows: -10)
    for r_ in range(n_in_rows):
        row_flushed[r_] = -10
    next_row = n_in_rows
    singles_slot = -1                         # cache slot of the row collecting values that belong to no atom
    atom_slots = []                           # cache slots of open atom rows (with a free half or pending members)
    max_pinned = max(2, cache // 2)
    prod = {}                                 # gate node -> vrow of its record
    saved_at = {}                             # ring value -> vrow of its SAVE lane
    lane_of = {}                              # ring values: node -> lane
    cmd_loads, cmd_flush = [], []             # per batch
    vrows = []                                # per vrow: list of (node, a_off, b_off, c_off, dst_off)
    stat = {"loads": 0, "flushes": 0, "atoms_kept": 0, "atoms_split": 0, "load_stall_ops": 0}
    t = 0
    placed = 0
    total = len(ops)

    def batch_of(v):
        return v // BATCH

    def ensure_batch(b):
        while len(cmd_loads) <= b:
            cmd_loads.append([])
            cmd_flush.append([])

    row_refs = {}                             # row -> consumer edges of its values that are not placed yet
    for x, n_ in input_refs.items():
        row_refs[home[x] // 64] = row_refs.get(home[x] // 64, 0) + n_

    def take_slot(now):
        """a cache slot whose content may be replaced: FREE, else a CLEAN one - rows no unplaced operation reads first,
        then the least recently used"""
        best, best_key = -1, None
        for s_ in range(cache):
            st = slot_state[s_]
            if st == FREE:
                return s_
            if st == CLEAN and slot_last[s_] <= now and slot_ready[s_] <= now and slot_free_at[s_] <= now:
                key = (1 if row_refs.get(slot_row[s_], 0) > 0 else 0, slot_last[s_])
                if best_key is None or key < best_key:
                    best, best_key = s_, key
        if best >= 0:
            del resident[slot_row[best]]
            slot_state[best] = FREE
        return best

    def open_row(now):
        nonlocal next_row
        s_ = take_slot(now)
        if s_ < 0:
            return -1
        row = next_row
        next_row += 1
        slot_state[s_] = OPEN
        slot_row[s_] = row
        slot_last[s_] = now
        slot_ready[s_] = 0
        slot_pend[s_] = 0
        slot_used[s_] = 0
        slot_open_at[s_] = now
        slot_dirty[s_] = True
        resident[row] = s_
        return s_

    def request_load(row, now):
        """returns the vrow from which the row is readable, or -1 if the request cannot be issued in this batch"""
        b = batch_of(now)
        ensure_batch(b)
        if len(cmd_loads[b]) >= MAX_LOADS or row_flushed.get(row, 1 << 60) > b - 2:
            return -1
        s_ = take_slot(now)
        if s_ < 0:
            return -1
        cmd_loads[b].append((row, s_))
        stat["loads"] += 1
        slot_state[s_] = LOADING
        slot_row[s_] = row
        slot_ready[s_] = (b + LOAD_DELAY) * BATCH
        slot_last[s_] = slot_ready[s_]
        slot_pend[s_] = 0
        slot_dirty[s_] = False
        resident[row] = s_
        return slot_ready[s_]

    def assign_home(nid, now):
        """bit-table slot for a stored value that is produced now; -1 = no cache slot can be opened right now"""
        nonlocal singles_slot
        at = atom_of.get(nid)
        if at is not None:
            atom = atoms[at[0]]
            if atom[0] is None:
                # reserve half a row for the whole atom
                s_ = -1
                for cand in atom_slots:
                    if slot_used[cand] <= 64 - ATOM:
                        s_ = cand
                        break
                if s_ < 0 and sum(1 for x in range(cache) if slot_state[x] == OPEN and slot_pend[x] > 0) < max_pinned:
                    s_ = open_row(now)
                    if s_ >= 0:
                        atom_slots.append(s_)
                if s_ < 0:
                    # no room to keep the word together: its members become ordinary values
                    for n_ in atom[1]:
                        atom_of.pop(n_, None)
                    stat["atoms_split"] += 1
                    at = None
                else:
                    atom[0] = slot_row[s_] * 64 + slot_used[s_]
                    slot_used[s_] += ATOM
                    slot_pend[s_] += ATOM
                    stat["atoms_kept"] += 1
                    if slot_used[s_] >= 64:
                        atom_slots.remove(s_)
            if at is not None:
                sl = atom[0] + at[1]
                s_ = resident[sl // 64]
                slot_pend[s_] -= 1
                return sl
        if singles_slot < 0 or slot_used[singles_slot] >= 64:
            singles_slot = open_row(now)
            if singles_slot < 0:
                return -1
        sl = slot_row[singles_slot] * 64 + slot_used[singles_slot]
        slot_used[singles_slot] += 1
        return sl

    def end_of_batch(b, final=False):
        """flush the rows that are complete (all handed-out positions produced; full, or old enough / final)"""
        nonlocal singles_slot
        ensure_batch(b)
        now = (b + 1) * BATCH - 1
        for s_ in range(cache):
            if slot_state[s_] != OPEN or slot_pend[s_] > 0 or len(cmd_flush[b]) >= MAX_FLUSH:
                continue
            full = slot_used[s_] >= 64
            if not (full or final or (now - slot_open_at[s_] >= CLOSE_AGE and s_ != singles_slot)):
                continue
            if slot_used[s_] == 0:
                slot_state[s_] = FREE
                del resident[slot_row[s_]]
            else:
                cmd_flush[b].append((slot_row[s_], s_))
                stat["flushes"] += 1
                row_flushed[slot_row[s_]] = b
                slot_free_at[s_] = (b + 2) * BATCH
                slot_state[s_] = CLEAN
                slot_dirty[s_] = False
                slot_ready[s_] = 0
            if s_ in atom_slots:
                atom_slots.remove(s_)
            if s_ == singles_slot:
                singles_slot = -1

    while placed < total:
        if t % BATCH == 0 and t:
            end_of_batch(batch_of(t) - 1)
        for item in buckets.pop(t, ()):
            heapq.heappush(heap, item)
        for s_ in range(cache):
            if slot_state[s_] == LOADING and slot_ready[s_] <= t:
                slot_state[s_] = CLEAN
        lanes = []
        produced = []
        deferred = []
        for x in expiry.pop(t, ()):
            if remaining.get(x, 0) <= 0 or x in home:
                continue
            sl = assign_home(x, t)
            if sl < 0 or len(lanes) >= 64:
                if t - prod[x] >= ring - 1 - LATENCY:
                    raise AssertionError("bit scheduler: no cache slot for a value that leaves the ring (cache too small)")
                expiry.setdefault(t + 1, []).append(x)
                continue
            s_ = resident[sl // 64]
            slot_last[s_] = t
            # copy lane: (x ^ 0) ^ 0 into the home entry; ring readers stay valid until the ring entry is reused
            lanes.append((-1, (prod[x] % ring) * 512 + lane_of[x] * 8, const_off, const_off, ring_bytes + s_ * 512 + (sl % 64) * 8))
            home[x] = sl
            row_refs[sl // 64] = row_refs.get(sl // 64, 0) + remaining.get(x, 0)
            saved_at[x] = t
            saved.add(x)
        while heap and len(lanes) < 64:
            if heap[0][0] > frontier + WINDOW and (produced or any(k > t for k in buckets)):
                break                      # everything that is ready belongs to the future
            item = heapq.heappop(heap)
            nid = item[2]
            ok = True
            retry = t + 1
            offs = []
            touched = []
            for x in (A[nid], B[nid], C[nid]):
                if x <= 1:
                    offs.append(const_off + 8 * x)
                    continue
                sv = saved_at.get(x)
                if sv is not None and t - sv < LATENCY:
                    # saved a moment ago: the home entry is not readable yet, the ring entry still is
                    offs.append((prod[x] % ring) * 512 + lane_of[x] * 8)
                    continue
                if x in home:
                    sl = home[x]
                    row = sl // 64
                    s_ = resident.get(row)
                    if s_ is None:
                        ok = False
                        ready = request_load(row, t)
                        retry = max(retry, ready if ready >= 0 else (batch_of(t) + 1) * BATCH)
                        stat["load_stall_ops"] += 1
                        continue
                    if slot_ready[s_] > t:
                        ok = False
                        retry = max(retry, slot_ready[s_])
                        continue
                    p_ = prod.get(x) if sv is None else None
                    if p_ is not None and t - p_ < LATENCY:
                        ok = False
                        retry = max(retry, p_ + LATENCY)
                        continue
                    offs.append(ring_bytes + s_ * 512 + (sl % 64) * 8)
                    touched.append(s_)
                else:
                    p_ = prod[x]
                    if t - p_ < LATENCY:
                        ok = False
                        retry = max(retry, p_ + LATENCY)
                        continue
                    assert t - p_ <= ring - 1, "ring value outlived the ring"
                    offs.append((p_ % ring) * 512 + lane_of[x] * 8)
            if ok and stored[nid]:
                sl = assign_home(nid, t)
                if sl < 0:
                    ok = False
                    retry = max(retry, t + 1)
                else:
                    home[nid] = sl
                    row_refs[sl // 64] = row_refs.get(sl // 64, 0) + remaining.get(nid, 0)
                    s_ = resident[sl // 64]
                    touched.append(s_)
                    dst = ring_bytes + s_ * 512 + (sl % 64) * 8
            if not ok:
                deferred.append((retry, item))
                continue
            for s_ in touched:
                if slot_last[s_] < t:
                    slot_last[s_] = t
            lane = len(lanes)
            if not stored[nid]:
                dst = (t % ring) * 512 + lane * 8
                lane_of[nid] = lane
            lanes.append((nid, offs[0], offs[1], offs[2], dst))
            produced.append(nid)
            for x in {A[nid], B[nid], C[nid]}:
                if x in remaining:
                    remaining[x] -= 1
                if x > 1 and x in home:
                    row_refs[home[x] // 64] -= 1
        for retry, item in deferred:
            buckets.setdefault(retry, []).append(item)
        for nid in produced:
            prod[nid] = t
            if target[nid] > frontier and target[nid] - level[nid] <= 2:
                frontier = target[nid]          # only operations without slack move the frontier (the others would let it creep)
            if not stored[nid] and remaining.get(nid, 0) > 0:
                expiry.setdefault(t + SAVE_AT, []).append(nid)
            for ci in consumers.get(nid, ()):
                pending[ci] -= 1
                if pending[ci] == 0:
                    buckets.setdefault(t + LATENCY, []).append((target[ci], -height[ci], ci))
                if pending[ci] <= PREFETCH_PENDING:
                    # the consumer is about to become ready: rows of its other operands should be on their way
                    for x in (A[ci], B[ci], C[ci]):
                        if x > 1 and x in home and (home[x] // 64) not in resident:
                            request_load(home[x] // 64, t)
        placed += len(produced)
        vrows.append(lanes)
        if _DEBUG and t % 400 == 0:
            tg = [target[n] for n in produced]
            print("[bitsched] t %6d frontier %5d placed %7d lanes %2d targets %s..%s loads %d saves %d heap %d open %d" % (
                t, frontier, placed, len(lanes), min(tg) if tg else None, max(tg) if tg else None, stat["loads"], len(saved), len(heap),
                sum(1 for x in range(cache) if slot_state[x] == OPEN)))
        t += 1
        if not heap and placed < total and not any(k >= t for k in buckets) and not any(k >= t for k in expiry):
            raise AssertionError("bit scheduler stalled")
    # ---- epilogue: pad to whole batches, flush what is still open ------------------------------------------------------
    while len(vrows) % BATCH:
        vrows.append([])
    b = len(vrows) // BATCH - 1
    end_of_batch(b, final=True)
    while any(slot_state[s_] == OPEN for s_ in range(cache)):
        for _ in range(BATCH):
            vrows.append([])
        b += 1
        end_of_batch(b, final=True)
    n_vrows = len(vrows)
    n_batches = n_vrows // BATCH
    ensure_batch(n_batches - 1)
    n_rows = next_row
    n_slots = n_rows * 64

    sig_slot = np.zeros(n_signals, dtype=np.uint32)
    for s_ in range(n_signals):
        sig_slot[s_] = home[int(sig_node[s_])]
    recs = np.zeros((n_vrows * 64, 2), dtype=np.uint32)
    c0 = const_off
    empty0 = c0 | (c0 << 16)
    for v, lanes in enumerate(vrows):
        base = v * 64
        for lane in range(64):
            if lane < len(lanes):
                nid, ao, bo, co, dst = lanes[lane]
                k = kind[nid] if nid >= 0 else 0
                recs[base + lane, 0] = ao | (1 if k & K_AND else 0) | (2 if k & K_OR else 0) | (bo << 16)
                recs[base + lane, 1] = co | (dst << 16)
            else:                          # idle lane: 0 ^ 0 ^ 0 into its own ring entry
                recs[base + lane, 0] = empty0
                recs[base + lane, 1] = c0 | (((v % ring) * 512 + lane * 8) << 16)
    cmds = np.zeros((n_batches, CMD_WORDS), dtype=np.uint32)
    for b in range(n_batches):
        ld, fl = cmd_loads[b], cmd_flush[b]
        assert len(ld) <= MAX_LOADS and len(fl) <= MAX_FLUSH
        cmds[b, 0] = len(ld) | (len(fl) << 8)
        for j, (row, s_) in enumerate(ld):
            cmds[b, 2 + 2 * j] = row * 512
            cmds[b, 3 + 2 * j] = ring_bytes + s_ * 512
        for j, (row, s_) in enumerate(fl):
            cmds[b, 2 + 2 * MAX_LOADS + 2 * j] = row * 512
            cmds[b, 3 + 2 * MAX_LOADS + 2 * j] = ring_bytes + s_ * 512
    bt = BitTape()
    bt.ring = ring
    bt.cache = cache
    bt.n_slots = n_slots
    bt.n_vrows = n_vrows
    bt.recs = recs
    bt.cmds = cmds
    bt.sig_slot = sig_slot
    bt.assert_slots = np.asarray([home[a] for a in net.asserts], dtype=np.uint32)
    bt.n_signals = n_signals
    bt.n_inputs = n_in
    bt.input_start = fc.main_input_start
    n_stored = sum(1 for nid in ops if stored[nid])
    bt.stats = {"vrows": n_vrows, "ops": total, "gates": total, "fill": total / (64.0 * max(1, t)), "slots": n_slots,
                "rows": n_rows, "stored_values": n_stored, "temp_values": len(extra_homes), "ring": ring, "cache": cache,
                "depth": net.stats.get("depth"), "lut_gates": net.stats.get("lut_gates"), "lut_depth": net.stats.get("lut_depth"),
                "asserts": len(net.asserts), "row_loads": stat["loads"], "row_flushes": stat["flushes"],
                "atoms_kept": stat["atoms_kept"], "atoms_split": stat["atoms_split"], "program_vrows": t,
                "save_lanes": len(saved)}
    bt.saved = saved
    return bt
