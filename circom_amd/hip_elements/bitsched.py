"""hip_elements bit-plane lowering, part 2: gate network (bitblast.BitNet) -> the program of `cw_bits_eval_kernel`.

Execution model (csrc/cw_bits.hip).  One wave evaluates the whole network for ONE group of instances; a value is a
mask (bit i = the value in instance i of the group).  The program is a sequence of VROWS; in a vrow every LANE does
one of
  * GATE   any 3-input boolean function (8-bit truth table) of three masks read from the wave's LDS RING,
  * LOAD   a mask read from the group's BIT TABLE in HBM (main inputs; values that left the ring),
and writes the result mask to ring entry (vrow mod R, lane) and, if the value is a signal (or must be re-loaded later),
to ITS slot of the bit table `T[group][slot]`.  Ring operands of vrow v+1 are requested while vrow v computes, so a
consumer sits at least TWO vrows after its producer and at most R-1.  Vector memory is touched per BATCH of 8 vrows
only: the records of batch b + 2 and the bit-table values of the LOAD lanes of batch b + 1 are requested when batch b
starts, the results of batch b are stored when it ends (so a LOAD lane of batch b reads what batch b - 2 or an older
one stored).  No barriers, no data-dependent branches: one wave, in-order LDS, in-order vector memory.

Signals that are copies of one another (`a.in <== b.out`: ~85 % of the signals of a circomlib circuit at --O0) are
the SAME value of the network: they share one slot.  `sig_slot[s]` maps every signal to its slot (egress and the
R1CS check read through it); a copy costs nothing at run time.
Bit-table slots: 0 = constant 0, 1 = constant 1 (all ones), 2 = reserved, main input k = IN_BASE + k, then one slot per
distinct signal value in the order the program stores them, then temps.

Record (4 x u32 per lane): a_off | b_off << 16,  c_off | tt << 16 | flags << 24,  g_off (LOAD lanes, else NONE),
d_off (byte offset of the destination slot, NONE = 0xFFFFFFFF: dropped by the buffer bounds check).

Scheduling: list scheduling into vrows of 64 lanes, priority = longest path to a sink; a value whose ring entry has
expired (or a main input) is re-loaded by a LOAD lane that the scheduler inserts on demand.  The vrow count
approaches max(2 * depth, operations / 64).
"""
from __future__ import annotations

import heapq
import os

import numpy as np

from .bitblast import BitNet

IN_BASE = 3                   # slot of main input 0
NONE = 0xFFFFFFFF
F_ASSERT = 1
DEFAULT_RING = 64
LATENCY = 2
LOAD_EVERY = 2                # LOAD lanes only sit in vrows t % LOAD_EVERY == 0: the kernel requests bit-table values for even
                              # vrows only (cw_bits.hip); measured on Sha256(2048): 16 932 -> 15 931 vrows, 210 K -> 105 K loads
BATCH = 8                     # vrows per batch of the kernel (memory traffic is issued per batch, cw_bits.hip)


class BitTape:
    def __init__(self):
        self.ring = DEFAULT_RING        # R: LDS ring entries (vrows); LDS bytes = R * 512
        self.n_slots = 0                # bit-table slots per group
        self.n_vrows = 0
        self.recs = None                # uint32 [n_vrows * 64, 4]
        self.sig_slot = None            # uint32 [n_signals]: slot holding each signal
        self.n_signals = 0
        self.n_inputs = 0
        self.input_start = 0
        self.stats = {}


def lower_bits(net: BitNet, fc, ring: int = DEFAULT_RING) -> BitTape:
    assert 4 * BATCH <= ring <= 128 and ring & (ring - 1) == 0
    n_nodes = len(net.tt)
    tt, A, B, C = net.tt, net.a, net.b, net.c
    n_signals = fc.n_signals
    sig_node = net.sig_node
    input_k = {nid: s - fc.main_input_start for s, nid in net.input_node.items()}
    is_gate = [0 <= t <= 0xFF and i > 1 for i, t in enumerate(tt)]
    is_signal = np.zeros(n_nodes, dtype=bool)
    is_signal[sig_node] = True
    # gates that no signal and no assertion depends on (helper computations of the witness code that only fed a
    # constraint) are not evaluated
    live = is_signal.tolist()
    for a in net.asserts:
        live[a] = True
    for nid in range(n_nodes - 1, 1, -1):
        if live[nid] and is_gate[nid]:
            live[A[nid]] = live[B[nid]] = live[C[nid]] = True
    is_gate = [g and live[i] for i, g in enumerate(is_gate)]

    # ---- operations: 'G' gate, 'A' assertion gate, 'L' load (created on demand) ------------------------------------
    o_kind, o_node, o_tt, o_src, o_pri = [], [], [], [], []

    def new_op(kind, node, t8, src, pri):
        o_kind.append(kind); o_node.append(node); o_tt.append(t8); o_src.append(src); o_pri.append(pri)
        return len(o_kind) - 1

    height = [0] * n_nodes
    for nid in range(n_nodes - 1, 1, -1):
        if not is_gate[nid]:
            continue
        h = height[nid] + 1
        for o in (A[nid], B[nid], C[nid]):
            if is_gate[o] and height[o] < h:
                height[o] = h
    prim = {}
    for nid in range(2, n_nodes):
        if is_gate[nid]:
            prim[nid] = new_op('G', nid, tt[nid], [o for o in (A[nid], B[nid], C[nid]) if o > 1], height[nid] + 1.0)
    for a in net.asserts:
        new_op('A', -1, 0xAA, [a], 0.5)
    n_static = len(o_kind)

    # ---- list scheduling with on-demand loads ---------------------------------------------------------------------------
    consumers = [[] for _ in range(n_nodes)]
    pending = [0] * n_static
    for oi in range(n_static):
        ops = {x for x in o_src[oi] if is_gate[x]}
        pending[oi] = len(ops)
        for x in ops:
            consumers[x].append(oi)
    heap = []
    seq = 0
    for oi in range(n_static):
        if pending[oi] == 0:
            heap.append((-o_pri[oi], seq, oi))
            seq += 1
    heapq.heapify(heap)
    buckets = {}
    copy_slot = {}                      # node -> vrow of its freshest ring copy
    copy_op = {}                        # node -> op that wrote that copy
    src_ops = {}                        # op -> ops whose ring entries it reads
    need_home = set()                   # gates that are no signal but must be stored (re-loaded later)
    prod_slot = {}                      # gate node -> vrow of its record
    vrows = []
    placed = 0
    total = n_static
    n_loads = 0
    carry_loads = []                    # loads that did not fit the vrow that asked for them
    t = 0

    def place_load(x, lanes, loading_now):
        nonlocal total, n_loads
        li = new_op('L', x, 0, [], 0.0)
        total += 1
        lanes.append(li)
        copy_slot[x] = t
        copy_op[x] = li
        loading_now.add(x)
        n_loads += 1
        if is_gate[x] and not is_signal[x]:
            need_home.add(x)

    while placed < total:
        for item in buckets.pop(t, ()):
            heapq.heappush(heap, item)
        lanes = []
        loading_now = set()
        load_row = t % LOAD_EVERY == 0          # LOAD lanes only sit in every LOAD_EVERY-th vrow (the kernel requests bit-table
                                                # values for those vrows only: fewer, denser vector-memory instructions)
        if load_row:
            for x in carry_loads[:64]:  # loads waiting for a load vrow go first
                place_load(x, lanes, loading_now)
            carry_loads = carry_loads[64:]
        carry_set = set(carry_loads)
        produced = []
        while heap and len(lanes) < 64:
            item = heapq.heappop(heap)
            oi = item[2]
            ok = True
            retry = t + 1
            for x in o_src[oi]:
                c = copy_slot.get(x)
                if c is None or t - c > ring - 1:
                    ok = False
                    if x in loading_now:
                        pass
                    elif is_gate[x] and t // BATCH - prod_slot[x] // BATCH < 2:
                        retry = max(retry, (prod_slot[x] // BATCH + 2) * BATCH)     # its store is not old enough yet
                    elif load_row and len(lanes) < 63:
                        place_load(x, lanes, loading_now)
                    elif x not in carry_set:
                        carry_loads.append(x)
                        carry_set.add(x)
                        if is_gate[x] and not is_signal[x]:
                            need_home.add(x)
                    if x in carry_set:
                        nxt_row = (t // LOAD_EVERY + 1) * LOAD_EVERY
                        retry = max(retry, nxt_row + LATENCY + (len(carry_loads) // 64) * LOAD_EVERY)
                    retry = max(retry, t + LATENCY)
                elif t - c < LATENCY:
                    ok = False
                    retry = max(retry, c + LATENCY)
            if not ok:
                # operands that would have left the ring by the time the op is retried are re-loaded now as well
                # (otherwise two operands can keep expiring in turn); the LOAD lanes of a batch read the bit table at the
                # start of the batch before, so the value must have been stored by the batch before that one
                for x in o_src[oi]:
                    c = copy_slot.get(x)
                    if (c is not None and retry - c > ring - 1 and x not in loading_now and x not in carry_set
                            and t // BATCH - prod_slot.get(x, -1 << 30) // BATCH >= 2):
                        if load_row and len(lanes) < 63:
                            place_load(x, lanes, loading_now)
                        else:
                            carry_loads.append(x)
                            carry_set.add(x)
                            if is_gate[x] and not is_signal[x]:
                                need_home.add(x)
                buckets.setdefault(retry, []).append(item)
                continue
            lanes.append(oi)
            src_ops[oi] = [copy_op[x] for x in o_src[oi]]
            if o_kind[oi] == 'G':
                produced.append(oi)
        for oi in produced:
            x = o_node[oi]
            prod_slot[x] = t
            copy_slot[x] = t
            copy_op[x] = oi
            for ci in consumers[x]:
                pending[ci] -= 1
                if pending[ci] == 0:
                    buckets.setdefault(t + LATENCY, []).append((-o_pri[ci], seq, ci))
                    seq += 1
        placed += len(lanes)
        vrows.append(lanes)
        t += 1
        if not heap and not carry_loads and placed < total and not any(k >= t for k in buckets):
            raise AssertionError("scheduler stalled")
    n_vrows = len(vrows)

    # ---- slots: constants, inputs, then stored values in program order (neighbouring lanes -> neighbouring slots) ----
    n_in = fc.n_main_inputs
    slot_of_node = {0: 0, 1: 1}
    for nid, k in input_k.items():
        slot_of_node[nid] = IN_BASE + k
    n_slots = IN_BASE + n_in
    slot_of_op = {}
    lane_of_op = {}
    first_sig = {}
    for s_ in range(n_signals - 1, -1, -1):
        first_sig[int(sig_node[s_])] = s_
    by_signal = os.environ.get("CW_BITS_SLOTS", "signal") == "signal"
    stored_all = []
    for v, lanes in enumerate(vrows):
        stored = [oi for oi in lanes if o_kind[oi] == 'G' and (is_signal[o_node[oi]] or o_node[oi] in need_home)]
        rest = [oi for oi in lanes if not (o_kind[oi] == 'G' and (is_signal[o_node[oi]] or o_node[oi] in need_home))]
        stored.sort(key=lambda oi: first_sig.get(o_node[oi], n_signals + o_node[oi]))
        lanes[:] = stored + rest
        for lane, oi in enumerate(lanes):
            slot_of_op[oi] = v
            lane_of_op[oi] = lane
        stored_all.extend(stored)
    if by_signal:
        # slots in the order of the FIRST SIGNAL each stored value is: the bits of a word (consecutive signals of a
        # component array) sit in consecutive slots, which is what the R1CS check reads together (a 32-bit word of a
        # BinSum row = one 64-byte scalar load per 8 bits); values that are no signal (re-load temps) follow
        stored_all.sort(key=lambda oi: first_sig.get(o_node[oi], n_signals + o_node[oi]))
    # (otherwise: in program order — neighbouring lanes of a vrow store to neighbouring slots)
    for oi in stored_all:
        slot_of_node[o_node[oi]] = n_slots
        n_slots += 1
    sig_slot = np.zeros(n_signals, dtype=np.uint32)
    for s_ in range(n_signals):
        sig_slot[s_] = slot_of_node[int(sig_node[s_])]

    out = np.zeros((n_vrows * 64, 4), dtype=np.uint32)
    out[:, 2:] = NONE
    for v, lanes in enumerate(vrows):
        for lane, oi in enumerate(lanes):
            row = out[v * 64 + lane]
            kind = o_kind[oi]
            offs = [0, 0, 0]
            for j, po in enumerate(src_ops.get(oi, ())):
                pv = slot_of_op[po]
                assert LATENCY <= v - pv <= ring - 1
                offs[j] = (pv % ring) * 512 + lane_of_op[po] * 8
            row[0] = offs[0] | (offs[1] << 16)
            row[1] = offs[2] | (o_tt[oi] << 16) | ((F_ASSERT if kind == 'A' else 0) << 24)
            if kind == 'L':
                row[2] = slot_of_node[o_node[oi]] * 8
            elif kind == 'G' and o_node[oi] in slot_of_node:
                row[3] = slot_of_node[o_node[oi]] * 8
    bt = BitTape()
    bt.ring = ring
    bt.n_slots = n_slots
    bt.n_vrows = n_vrows
    bt.recs = out
    bt.sig_slot = sig_slot
    bt.n_signals = n_signals
    bt.n_inputs = n_in
    bt.input_start = fc.main_input_start
    kinds = {k: o_kind.count(k) for k in "GAL"}
    bt.stats = {"vrows": n_vrows, "ops": total, "gates": kinds['G'], "loads": kinds['L'], "fill": total / (64.0 * n_vrows),
                "slots": n_slots, "stored_values": n_slots - IN_BASE - n_in, "temp_values": len(need_home),
                "ring": ring, "depth": net.stats.get("depth"), "asserts": len(net.asserts)}
    return bt
