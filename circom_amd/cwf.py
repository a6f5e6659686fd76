"""`.cwf` - the FLAT CIRCUIT interchange file between a circom front-end and the hip_elements back-end.

The north-star keeps circom's Rust front-end (parser, type analysis, constraint generation, DAG, IR buckets) and adds a
`hip_elements` code producer.  That producer's job ends where `circom_amd/frontend/flatten.py` ends: a single traced
execution of the witness program over GLOBAL signal ids (component tree unrolled, loops over knowns unrolled, input-counter
firing order resolved - SURVEY Appendix B), the constant list, the main component's shape, the constraints.  This module
fixes that hand-over as a file, so that the Rust side (integration/code_producers/src/hip_elements/, unbuildable here: no
cargo) and the Python stand-in emit the same bytes, and `python -m circom_amd.hip_backend x.cwf` lowers either.

Layout (little endian):
     0  "CWFL" | u32 version = 2 | u32 n64 (limbs of the prime) | u32 flags (0)
    16  prime, n64 * 8 bytes
        10 x u32: n_signals, n_temps, n_constants, main_input_start, n_main_inputs, n_public_inputs, n_outputs,
                  n_input_names, n_ops, n_constraints
        constants        n_constants x n64*8 bytes (canonical residues)
        input names      per name: u32 len | bytes | u32 first signal | u32 size
        flat code        9 columns x n_ops x i64: op, dk, dv, ak, av, bk, bv, ck, cv
                         (opcodes: circom_amd/opcodes.py = the OperatorType of compute_bucket.rs:7-34 plus COPY / SELECT /
                          ASSERT_* / CALL; operand kinds: 0 signal, 1 temporary, 2 constant index, 3 none)
        constraints      per constraint, per part A, B, C: u32 n_terms | n_terms x (u32 signal | n64*8 bytes coefficient)
                         (A * B - C = 0 over signal ids, signal 0 = the constant 1: circom_algebra/src/algebra.rs:998-1009)
        functions        u32 n | per function: u32 len | name | u32 n_regs | u32 n_args | u32 n_ret | u32 ret_base | u32 n_ins |
                         u32 n_fconsts | n_ins x 6 x i64 {op, dst or -1, a kind, a, b kind, b} (kind 0 none, 1 register, 2 index
                         into the function's constants) | n_fconsts x n64*8 bytes
                         (register bytecode of circom functions with run-time control flow, frontend/rtcode.py)
        io map           u32 n | per template: u32 id | u32 n_defs | per def: u32 offset | u32 n_dims | dims | u32 size | u32 bus
        log strings      u32 n | per string: u32 len | bytes      (string table of the LOG rows: opcodes.py LOG)
"""
from __future__ import annotations

import struct
from types import SimpleNamespace

import numpy as np

from .field import Fp, PRIMES

MAGIC = b"CWFL"
VERSION = 2
COLS = ("op", "dk", "dv", "ak", "av", "bk", "bv", "ck", "cv")


def write_cwf(path, fc):
    q = fc.fp.q
    n64 = (q.bit_length() + 63) // 64
    nb = 8 * n64
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<III", VERSION, n64, 0))
        f.write(q.to_bytes(nb, "little"))
        n_ops = len(fc.code["op"])
        f.write(struct.pack("<10I", fc.n_signals, fc.n_temps, len(fc.constants), fc.main_input_start, fc.n_main_inputs,
                            fc.n_pub_in, fc.n_outputs, len(fc.inputs), n_ops, len(fc.constraints)))
        f.write(b"".join(int(c).to_bytes(nb, "little") for c in fc.constants))
        for name, start, size in fc.inputs:
            b = name.encode()
            f.write(struct.pack("<I", len(b)) + b + struct.pack("<II", start, size))
        for col in COLS:
            f.write(np.ascontiguousarray(fc.code[col], dtype="<i8").tobytes())
        for cons in fc.constraints:
            for part in cons:
                f.write(struct.pack("<I", len(part)))
                for sig in sorted(part):
                    f.write(struct.pack("<I", sig) + int(part[sig]).to_bytes(nb, "little"))
        fns = list(getattr(fc, "functions", ()))
        f.write(struct.pack("<I", len(fns)))
        for fn in fns:
            fconsts, rows = [], []

            def opnd(x):
                if x is None:
                    return 0, 0
                if x[0] == 'r':
                    return 1, int(x[1])
                fconsts.append(int(x[1]))
                return 2, len(fconsts) - 1

            for op, d, a, b_ in fn["code"]:
                ak, av = opnd(a)
                bk, bv = opnd(b_)
                rows.append((int(op), -1 if d is None else int(d), ak, av, bk, bv))
            nm = fn.get("name", "").encode()
            f.write(struct.pack("<I", len(nm)) + nm)
            f.write(struct.pack("<6I", fn["n_regs"], fn["n_args"], fn["n_ret"], fn["ret_base"], len(rows), len(fconsts)))
            f.write(np.asarray(rows, dtype="<i8").reshape(-1, 6).tobytes())
            f.write(b"".join(c.to_bytes(nb, "little") for c in fconsts))
        io_map = list(getattr(fc, "io_map", ()))
        f.write(struct.pack("<I", len(io_map)))
        for tid, defs in io_map:
            f.write(struct.pack("<II", tid, len(defs)))
            for offset, dims, size, bus in defs:
                f.write(struct.pack("<II", offset, len(dims)) + b"".join(struct.pack("<I", d) for d in dims) + struct.pack("<II", size, bus))
        strings = list(getattr(fc, "log_strings", ()))
        f.write(struct.pack("<I", len(strings)))
        for t in strings:
            b = t.encode()
            f.write(struct.pack("<I", len(b)) + b)


def read_cwf(path):
    """A FlatCircuit-shaped object: everything hip_elements.lower / bitblast / writers.write_dat / write_r1cs read.
    (`.sym` names and the oracle's reference-style C++ need the component tree, which stays with the front-end.)"""
    b = open(path, "rb").read()
    if b[:4] != MAGIC:
        raise ValueError("not a .cwf file")
    version, n64, flags = struct.unpack_from("<III", b, 4)
    if version != VERSION or flags:
        raise ValueError("unsupported .cwf version")
    nb = 8 * n64
    off = 16
    q = int.from_bytes(b[off:off + nb], "little")
    off += nb
    (n_signals, n_temps, n_consts, main_in0, n_in, n_pub, n_out, n_names, n_ops, n_cons) = struct.unpack_from("<10I", b, off)
    off += 40
    consts = [int.from_bytes(b[off + i * nb: off + (i + 1) * nb], "little") for i in range(n_consts)]
    off += n_consts * nb
    inputs = []
    for _ in range(n_names):
        (ln,) = struct.unpack_from("<I", b, off)
        name = b[off + 4: off + 4 + ln].decode()
        start, size = struct.unpack_from("<II", b, off + 4 + ln)
        inputs.append((name, start, size))
        off += 12 + ln
    code = {}
    for col in COLS:
        code[col] = np.frombuffer(b, dtype="<i8", count=n_ops, offset=off).astype(np.int64)
        off += 8 * n_ops
    constraints = []
    for _ in range(n_cons):
        parts = []
        for _p in range(3):
            (nt,) = struct.unpack_from("<I", b, off)
            off += 4
            part = {}
            for _t in range(nt):
                (sig,) = struct.unpack_from("<I", b, off)
                part[sig] = int.from_bytes(b[off + 4: off + 4 + nb], "little")
                off += 4 + nb
            parts.append(part)
        constraints.append(tuple(parts))
    (n_fn,) = struct.unpack_from("<I", b, off)
    off += 4
    functions = []
    for _ in range(n_fn):
        (ln,) = struct.unpack_from("<I", b, off)
        fname = b[off + 4: off + 4 + ln].decode()
        off += 4 + ln
        n_regs, n_args, n_ret, ret_base, n_ins, n_fc = struct.unpack_from("<6I", b, off)
        off += 24
        rows = np.frombuffer(b, dtype="<i8", count=n_ins * 6, offset=off).reshape(n_ins, 6).tolist()
        off += 48 * n_ins
        fconsts = [int.from_bytes(b[off + i * nb: off + (i + 1) * nb], "little") for i in range(n_fc)]
        off += n_fc * nb

        def opnd(k, v):
            return None if k == 0 else ('r', v) if k == 1 else ('c', fconsts[v])

        fcode = [[op, None if d < 0 else d, opnd(ak, av), opnd(bk, bv)] for op, d, ak, av, bk, bv in rows]
        functions.append({"name": fname, "n_args": n_args, "n_ret": n_ret, "ret_base": ret_base, "n_regs": n_regs, "code": fcode,
                          "consts": consts})
    (n_io,) = struct.unpack_from("<I", b, off)
    off += 4
    io_map = []
    for _ in range(n_io):
        tid, nd = struct.unpack_from("<II", b, off)
        off += 8
        defs = []
        for _d in range(nd):
            offset, ndim = struct.unpack_from("<II", b, off)
            dims = struct.unpack_from("<%dI" % ndim, b, off + 8)
            size, bus = struct.unpack_from("<II", b, off + 8 + 4 * ndim)
            off += 16 + 4 * ndim
            defs.append((offset, tuple(dims), size, bus))
        io_map.append((tid, defs))
    (n_str,) = struct.unpack_from("<I", b, off)
    off += 4
    log_strings = []
    for _ in range(n_str):
        (ln,) = struct.unpack_from("<I", b, off)
        log_strings.append(b[off + 4: off + 4 + ln].decode())
        off += 4 + ln
    if off != len(b):
        raise ValueError(".cwf: trailing bytes")
    prime = next((n for n, p in PRIMES.items() if p == q), "")
    return SimpleNamespace(fp=Fp(q, prime), prime=prime, n_signals=n_signals, n_temps=n_temps, constants=consts,
                           main_input_start=main_in0, n_main_inputs=n_in, n_pub_in=n_pub, n_prv_in=n_in - n_pub, n_outputs=n_out,
                           inputs=inputs, code=code, constraints=constraints, functions=functions, io_map=io_map,
                           log_strings=log_strings)
