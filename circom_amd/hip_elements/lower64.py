"""hip_elements for the 64-bit runtime (`--prime goldilocks`): the flat witness program as one row per operation.

The reference has a second runtime for this prime: `code_producers/src/c_elements/goldilocks/fr.hpp` (a field element is a
plain uint64) with `common64/{main,calcwit}.cpp`, value-style emitted code and a `.dat` without a constants section
(c_code_generator.rs:838-841; constants are literals, value_bucket.rs:82-86).  The 256-bit engine's lowering is built on
"q is large" (reduction-free short products, lazy integer sums, bit extraction, Montgomery radix 2^261) - none of it applies,
and with one register per value none of it is needed: `csrc/cw64.hip` executes the flat code as it stands.

  row = 8 x u32: op | dk << 8 | ak << 10 | bk << 12 | ck << 14 (kind 0 = table slot, 2 = constant, 3 = none); dst; a; b; c;
                 index of the flat operation (what a failed check reports); 0; 0
  slots: signal s = slot s (slot 0 = the constant 1), temporary t = n_signals + t
"""
from __future__ import annotations

import struct

import numpy as np

from .. import opcodes as O
from .writers import hashmap_size, build_hash_map, dat_io_map

GOLDILOCKS = 18446744069414584321
MAGIC = b"CW64"
VERSION = 1


class Tape64:
    def __init__(self):
        self.rows = None            # uint32 [n_rows, 8]
        self.consts = None          # uint64 [n_consts]
        self.n_signals = self.n_slots = 0


def lower64(fc) -> Tape64:
    if fc.fp.q != GOLDILOCKS:
        raise ValueError("the 64-bit runtime serves the Goldilocks prime only")
    code = fc.code
    op = code["op"]
    if ((op == O.CALL) | (op == O.LOG)).any():
        raise ValueError("run-time functions and log() are not available in the 64-bit runtime yet")
    keep = op != O.RUN
    idx = np.nonzero(keep)[0]
    n = len(idx)
    rows = np.zeros((n, 8), dtype=np.uint32)
    ns = fc.n_signals

    def operand(kk, vv):
        k = code[kk][idx]
        v = code[vv][idx].astype(np.int64)
        kind = np.where(k == O.K_SIG, 0, np.where(k == O.K_TMP, 0, np.where(k == O.K_CONST, 2, 3)))
        val = np.where(k == O.K_TMP, v + ns, np.where(k == O.K_NONE, 0, v))
        return kind.astype(np.uint32), val.astype(np.uint32)

    dk, dv = operand("dk", "dv")
    ak, av = operand("ak", "av")
    bk, bv = operand("bk", "bv")
    ck, cv = operand("ck", "cv")
    no_dst = np.isin(op[idx], list(O.NO_DST))
    dk = np.where(no_dst, 3, dk)
    assert not (dk == 2).any(), "a constant cannot be a destination"
    rows[:, 0] = op[idx].astype(np.uint32) | (dk << 8) | (ak << 10) | (bk << 12) | (ck << 14)
    rows[:, 1] = np.where(dk == 0, dv, 0)
    rows[:, 2], rows[:, 3], rows[:, 4] = av, bv, cv
    rows[:, 5] = idx.astype(np.uint32)
    t = Tape64()
    t.rows = rows
    t.consts = np.asarray([int(c) % GOLDILOCKS for c in fc.constants], dtype=np.uint64)
    t.n_signals = ns
    t.n_slots = ns + max(int(fc.n_temps), 0)
    return t


def write_tape64(path, fc, t: Tape64):
    """layout: csrc/cw_host.cpp load_tape64"""
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<I", VERSION) + struct.pack("<Q", GOLDILOCKS))
        f.write(struct.pack("<12I", fc.n_signals, fc.n_signals, len(t.consts), fc.main_input_start, fc.n_main_inputs, len(fc.inputs),
                            hashmap_size(len(fc.inputs)), fc.n_pub_in, t.n_slots, len(t.rows), len(getattr(fc, "io_map", ())), 0))
        f.write(t.consts.astype("<u8").tobytes())
        f.write(np.arange(fc.n_signals, dtype="<u4").tobytes())
        for name, start, size in fc.inputs:
            b = name.encode()
            f.write(struct.pack("<I", len(b)) + b + struct.pack("<II", start, size))
        f.write(np.ascontiguousarray(t.rows, dtype="<u4").tobytes())


def write_dat64(path, fc):
    """`.dat` as common64/main.cpp reads it (hash map, witness list, io map - no constant table: the reference inlines constants
    as literals for this prime, value_bucket.rs:82-86)"""
    size = hashmap_size(len(fc.inputs))
    with open(path, "wb") as f:
        f.write(b"".join(struct.pack("<QQQ", *e) for e in build_hash_map(fc.inputs, size)))
        f.write(np.arange(fc.n_signals, dtype="<u8").tobytes())
        f.write(dat_io_map(getattr(fc, "io_map", ())))
    return size
