"""hip_elements bit-plane lowering, part 2: gate network (bitblast.BitNet) -> the program of `cw_bits_kernel`.

Execution model (csrc/cw_bits.hip).  One wave evaluates the whole network for ONE group of 64 instances; a value is a
64-bit mask (bit i = the value in instance i).  The program is a sequence of VROWS; in a vrow every lane evaluates one
3-input gate (any truth table) on three operand masks and writes the result mask
  * to the wave's LDS RING (entry `vrow mod R`, lane) — where the next R-1 vrows find it,
  * to up to four slots of the group's BIT TABLE in HBM (`T[group][slot]`, 8 bytes each): the signals this value is
    (a `x <== y` copy is one more destination, never a gate) and, if some consumer is R or more vrows away, a temp slot.
Operand kinds: PREV (lane j of the previous vrow's result, read with ds_bpermute), RING (entry written 2..R-1 vrows
ago), GLOBAL (bit-table slot: inputs, constants, values older than the ring).  Ring/global operands of vrow v+1 are
requested while vrow v computes; PREV operands after it.  There are no barriers: one wave, in-order LDS and in-order
vector memory make every read-after-write of the rules above safe by construction.

Bit-table slots: 0 = constant 0, 1 = constant 1 (all ones), 2 = reserved, signal s = SIG_BASE + s, then temps.

Scheduling: list scheduling of the gates into vrows of 64 (priority = longest path to a sink), so the vrow count
approaches max(depth, gates / 64).  Lanes of a vrow are sorted by their first destination slot (consecutive
signals of a component are written by neighbouring lanes: coalesced 8-byte stores).
"""
from __future__ import annotations

import heapq

import numpy as np

from .bitblast import BitNet

SIG_BASE = 3
SLOT_ZERO, SLOT_ONE, SLOT_RSV = 0, 1, 2
K_GLOBAL, K_RING, K_PREV = 0, 1, 2
F_ASSERT = 1 << 8
DEFAULT_RING = 128


class BitTape:
    def __init__(self):
        self.ring = DEFAULT_RING        # R: LDS ring entries (vrows); LDS bytes = R * 512
        self.n_slots = 0                # bit-table slots per group
        self.n_vrows = 0
        self.recs = None                # uint32 [n_vrows * 64, 8]: a, b, c, tt|flags, d0, d1, d2, d3
        self.n_signals = 0
        self.n_inputs = 0
        self.input_start = 0
        self.stats = {}


def lower_bits(net: BitNet, fc, ring: int = DEFAULT_RING) -> BitTape:
    n_nodes = len(net.tt)
    tt, A, B, C = net.tt, net.a, net.b, net.c
    n_signals = fc.n_signals
    sig_node = net.sig_node
    # ---- destinations of every node: the signals that alias it ---------------------------------------------------
    dests = [None] * n_nodes
    order = np.argsort(sig_node, kind="stable")
    sn_sorted = sig_node[order]
    starts = np.flatnonzero(np.r_[True, sn_sorted[1:] != sn_sorted[:-1]])
    ends = np.r_[starts[1:], len(order)]
    for s0, s1 in zip(starts.tolist(), ends.tolist()):
        dests[int(sn_sorted[s0])] = order[s0:s1].tolist()
    input_sig = {nid: s for s, nid in net.input_node.items()}

    # ---- records: one per gate (+ more for values with > 4 destinations, + copies of inputs / constants) ---------------
    # rec = [node (value computed), tt, a, b, c (operand NODES), dest slots list, flags, primary?]
    recs = []
    prim_of = {}                     # node -> index of its primary record

    def add_recs(node, t8, a, b, c, dlist, flags=0):
        first = True
        i = 0
        while first or i < len(dlist):
            r = [node, t8, a, b, c, dlist[i:i + 4], flags, first]
            if first:
                prim_of[node] = len(recs)
            recs.append(r)
            first = False
            i += 4

    for nid in range(n_nodes):
        d = dests[nid] or []
        t8 = tt[nid]
        if nid <= 1:
            # constants: slots 0/1 are initialised by the runtime; signals equal to a constant are written by gates
            if d:
                for i in range(0, len(d), 4):
                    recs.append([-1, 0xFF if nid else 0x00, 0, 0, 0, [SIG_BASE + s for s in d[i:i + 4]], 0, False])
            continue
        if t8 > 0xFF:
            # main input: its own slot is written by the ingest kernel; further aliases are copies (identity gates)
            own = input_sig[nid]
            rest = [s for s in d if s != own]
            for i in range(0, len(rest), 4):
                recs.append([-1, 0xAA, nid, 0, 0, [SIG_BASE + s for s in rest[i:i + 4]], 0, False])
            continue
        add_recs(nid, t8, A[nid], B[nid], C[nid], [SIG_BASE + s for s in d])
    for a in net.asserts:
        recs.append([-1, 0xAA, a, 0, 0, [], F_ASSERT, False])
    n_recs = len(recs)

    # ---- list scheduling --------------------------------------------------------------------------------------------
    is_gate = [0 <= t <= 0xFF and i > 1 for i, t in enumerate(tt)]
    height = [0] * n_nodes
    for nid in range(n_nodes - 1, 1, -1):
        if not is_gate[nid]:
            continue
        h = height[nid] + 1
        for o in (A[nid], B[nid], C[nid]):
            if is_gate[o] and height[o] < h:
                height[o] = h
    consumers = [[] for _ in range(n_nodes)]      # node -> records waiting for it
    pending = [0] * n_recs
    for ri, r in enumerate(recs):
        ops = {o for o in (r[2], r[3], r[4]) if is_gate[o]}
        pending[ri] = len(ops)
        for o in ops:
            consumers[o].append(ri)
    ready = []                                      # heap of (-priority, rec index)
    for ri, r in enumerate(recs):
        if pending[ri] == 0:
            heapq.heappush(ready, (-(height[r[0]] if r[0] >= 0 and r[7] else -1), ri))
    vrow_of_node = [-1] * n_nodes                  # gates: vrow of the primary record
    lane_of_node = [0] * n_nodes
    vrows = []
    done = 0
    while done < n_recs:
        take = []
        while ready and len(take) < 64:
            take.append(heapq.heappop(ready)[1])
        assert take, "scheduler stalled (cyclic network?)"
        v = len(vrows)
        # lanes sorted by first destination slot (coalesced stores); records without one keep their order at the end
        take.sort(key=lambda ri: (recs[ri][5][0] if recs[ri][5] else 1 << 40))
        vrows.append(take)
        newly = []
        for lane, ri in enumerate(take):
            r = recs[ri]
            if r[7]:
                vrow_of_node[r[0]] = v
                lane_of_node[r[0]] = lane
                newly.append(r[0])
        for nid in newly:
            for ci in consumers[nid]:
                pending[ci] -= 1
                if pending[ci] == 0:
                    rr = recs[ci]
                    heapq.heappush(ready, (-(height[rr[0]] if rr[0] >= 0 and rr[7] else -1), ci))
        done += len(take)
    n_vrows = len(vrows)

    # ---- operand kinds; which gates need a global temp slot --------------------------------------------------------------
    far = set()
    n_prev = n_ring = n_glob = 0
    for v, take in enumerate(vrows):
        for ri in take:
            r = recs[ri]
            for o in (r[2], r[3], r[4]):
                if is_gate[o] and v - vrow_of_node[o] >= ring and not dests[o]:
                    far.add(o)
    home = {}
    n_slots = SIG_BASE + n_signals
    for nid in sorted(far, key=lambda x: (vrow_of_node[x], lane_of_node[x])):
        home[nid] = n_slots
        n_slots += 1

    def node_slot(o):
        if o <= 1:
            return o
        if not is_gate[o]:
            return SIG_BASE + input_sig[o]
        d = dests[o]
        return SIG_BASE + d[0] if d else home[o]

    out = np.zeros((n_vrows * 64, 8), dtype=np.uint32)
    for v, take in enumerate(vrows):
        for lane, ri in enumerate(take):
            r = recs[ri]
            row = out[v * 64 + lane]
            for j, o in enumerate((r[2], r[3], r[4])):
                if is_gate[o]:
                    dist = v - vrow_of_node[o]
                    assert dist >= 1
                    if dist == 1:
                        row[j] = (K_PREV << 30) | (lane_of_node[o] * 4)
                        n_prev += 1
                    elif dist < ring:
                        row[j] = (K_RING << 30) | ((vrow_of_node[o] % ring) * 512 + lane_of_node[o] * 8)
                        n_ring += 1
                    else:
                        row[j] = (K_GLOBAL << 30) | (node_slot(o) * 8)
                        n_glob += 1
                else:
                    row[j] = (K_GLOBAL << 30) | (node_slot(o) * 8)
                    n_glob += 1 if o > 1 else 0
            row[3] = r[1] | r[6]
            dl = list(r[5])
            if r[7] and r[0] in home:
                dl.append(home[r[0]])
            assert len(dl) <= 4 or not r[7]
            if len(dl) > 4:                 # cannot happen: primary records carry <= 4 signal destinations + the temp
                raise AssertionError
            for j, s in enumerate(dl):
                row[4 + j] = s * 8
        # unused lanes: constant-0 gate without destination (all zero record)
    # a primary record with 4 signal destinations AND a far temp slot would need 5 entries: handled by giving the temp
    # slot to such nodes through an extra record is not needed — signal-aliased nodes use their signal slot as home.
    bt = BitTape()
    bt.ring = ring
    bt.n_slots = n_slots
    bt.n_vrows = n_vrows
    bt.recs = out
    bt.n_signals = n_signals
    bt.n_inputs = fc.n_main_inputs
    bt.input_start = fc.main_input_start
    bt.stats = {"vrows": n_vrows, "records": n_recs, "gates": int(sum(is_gate)), "fill": n_recs / (64.0 * n_vrows),
                "prev_operands": n_prev, "ring_operands": n_ring, "global_operands": n_glob,
                "temp_slots": n_slots - SIG_BASE - n_signals, "ring": ring, "depth": net.stats.get("depth"),
                "asserts": len(net.asserts)}
    return bt
