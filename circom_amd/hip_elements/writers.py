"""File emitters of the hip_elements back-end.

  write_dat   `<name>.dat`  byte-for-byte the reference layout (c_code_generator.rs:575-679,818-865;
              reader main.cpp:22-124): input hash map, witness->signal list, constant table.  The same
              file feeds the reference C++ runtime (oracle) and the HIP runtime.
  write_tape  `<name>.cwt`  the batched schedule for the HIP kernels (new; layout below).
  write_r1cs  `<name>.r1cs` iden3 binary R1CS exactly as constraint_writers/src/r1cs_writer.rs lays it
              out (section order 2,1,3 as dag/src/r1cs_porting.rs:13-46 writes it).
  write_sym   `<name>.sym`  (constraint_writers sym format, docs formats/sym.md).
  write_wtns  `.wtns` v2   (main.cpp:288-334 = witness_calculator.js:212-276).
"""
from __future__ import annotations

import struct

import numpy as np

from ..frontend.flatten import FlatCircuit
from .lower import Tape

FNV_OFFSET = 0xCBF29CE484222325
FNV_PRIME = 0x100000001B3
M64 = (1 << 64) - 1


def fnv1a(s: str) -> int:
    """64-bit FNV-1a (calcwit.cpp:17-24 == components/mod.rs:50-55)."""
    h = FNV_OFFSET
    for c in s.encode():
        h ^= c
        h = (h * FNV_PRIME) & M64
    return h


def hashmap_size(n_input_names: int) -> int:
    """c_elements/mod.rs:167-169: max(2^ceil(log2 n), 256)."""
    n = 1
    while n < n_input_names:
        n *= 2
    return max(n, 256)


def build_hash_map(inputs, size):
    """generate_hash_map, c_code_generator.rs:575-587 (linear probing, empty = signalid 0)."""
    tab = [(0, 0, 0)] * size
    for name, start, sz in inputs:
        h = fnv1a(name)
        p = h % size
        while tab[p][1] != 0:
            p = (p + 1) % size
        tab[p] = (h, start, sz)
    return tab


def dat_constant(v: int, fp) -> bytes:
    """One 40-byte FrElement of the constant table (c_code_generator.rs:616-679)."""
    q = fp.q
    n = v % q
    nn = n - q if n > q // 2 else n
    if -2147483648 <= nn <= 2147483647:
        head = struct.pack("<I", nn & 0xFFFFFFFF) + struct.pack("<I", 0x40000000)
    else:
        head = struct.pack("<I", 0) + struct.pack("<I", 0xC0000000)
    return head + ((n * fp.R) % q).to_bytes(8 * fp.n64, "little")


def write_dat(path, fc: FlatCircuit, witness2signal=None):
    size = hashmap_size(len(fc.inputs))
    tab = build_hash_map(fc.inputs, size)
    if witness2signal is None:
        witness2signal = np.arange(fc.n_signals, dtype=np.uint64)
    with open(path, "wb") as f:
        f.write(b"".join(struct.pack("<QQQ", *e) for e in tab))
        f.write(np.asarray(witness2signal, dtype="<u8").tobytes())
        f.write(b"".join(dat_constant(v, fc.fp) for v in fc.constants))
        f.write(dat_io_map(getattr(fc, "io_map", ())))
        f.write(dat_bus_field_map(getattr(fc, "bus_field_map", ())))
    return size


def dat_bus_field_map(buses) -> bytes:
    """bus-field map, the last section of the `.dat` (c_code_generator.rs:740-794 `generate_dat_bus_field_info`; reader
    main.cpp:95-121): per bus instance the number of fields and, per field, offset | number of dimensions - 1 (0 for a scalar)
    | dimensions[1..] | size of one element | id of the field's own bus (0 when it is a signal: the reference writes 0 for
    None) - all u32 little endian.  The reference binary reads get_size_of_bus_field_map() entries; cw_load reads to the
    end of the file.  Every access is resolved at trace time here, so - like the io-map - the section only matters for the
    files: the reference runtime loads a `.dat` of a circuit with buses, and the loader validates it."""
    out = []
    for fields in buses:
        out.append(struct.pack("<I", len(fields)))
        for offset, dims, size, bus, _name in fields:
            out.append(struct.pack("<II", offset, max(len(dims) - 1, 0)))
            out.extend(struct.pack("<I", d) for d in dims[1:])
            out.append(struct.pack("<II", size, bus or 0))
    return b"".join(out)


def dat_io_map(io_map) -> bytes:
    """io-map section of the `.dat` (c_code_generator.rs:681-738 `generate_dat_io_signals_info`; reader main.cpp:60-92):
    the template ids, then per template: number of io signals and, per signal, offset | number of lengths - 1 (0 for a
    scalar) | lengths[1..] | element size | bus id - all u32 little endian.  (The bus-field map follows: dat_bus_field_map.)"""
    out = [struct.pack("<I", tid) for tid, _ in io_map]
    for _, defs in io_map:
        out.append(struct.pack("<I", len(defs)))
        for offset, dims, size, bus in defs:
            out.append(struct.pack("<II", offset, max(len(dims) - 1, 0)))
            out.extend(struct.pack("<I", d) for d in dims[1:])
            out.append(struct.pack("<II", size, bus))
    return b"".join(out)


TAPE_MAGIC = b"CWTP"
TAPE_VERSION = 11


def write_tape(path, tapes, bittape=None, jit=None, fpjit=(), r1cs_id=None):
    """`.cwt` layout (little endian).  `tapes` = one Tape or a list of Tapes of the SAME circuit lowered with
    different strand counts (the runtime picks the variant that fills the chip for the batch at hand).
         0  "CWTP" | u32 version | u32 n64 | u32 n_variants
        16  prime, n64*8 bytes
            16 x u32: n_signals, n_witness, n_consts, main_input_start, n_main_inputs, n_input_names,
                      hashmap_size, rbits (Montgomery radix exponent of MMUL rows; bit 16 set = the value table holds
                      Montgomery forms, lower.py pass A6), n_lconsts, n_public_inputs,
                      n_bit_programs (0 | 1), n_functions, constants in the .dat, io-map templates in the .dat,
                      n_log_statements, n_log_values (hidden signals: the LAST n_log_values of n_signals hold the arguments
                      of the log statements, lower.py)
            consts          n_consts x n64*8 bytes (raw residues as the schedule expects them)
            lconsts         n_lconsts x n64*8 bytes (coef*R' of D_DOTC terms; the runtime keeps them as 29-bit limbs)
            witness2signal  n_witness x u32
            input names     per name  u32 len | bytes | u32 start | u32 size
            functions       n_functions x { u32 n_regs | u32 n_ins | u32 native kind (0 none, 1 mod_inv, 2 ec_add, 3 ec_double) |
                            u32 limb bits | u32 limbs | 32 bytes modulus | n_ins x 4 x u32 }   (device bytecode, lower.py D_CALL; the
                            native closed form of a pure big-integer function, circuits/bigint_func.py)
            log program     n_log_statements x { u32 flat operation that ends the statement | u32 n_items |
                            n_items x { u32 0 | u32 len | bytes   (a string)   or   u32 1 | u32 j   (the j-th logged value) } }
            per variant     u32 n_strands | u32 n_tslots | u32 n_rows | u32 n_extras | u32 n_lds | u32 n_terms | u32 kind | u32 shape
                            stream_off | extra_off | term_off      ((n_strands+1) x u32 each)
                            rows n_rows x 4 x u32 | extras n_extras x u32 | terms n_terms x 4 x u32
                            kind 0: shape = n_seq, then seq_off (n_strands+1) x u32 | seqs n_seq x u32: for every row that can
                            fail a check (ASSERT_EQ/NZ, IDIV, MOD, CALL), in stream order, the index of its flat operation
                            kind 1 (pipelined, pipe.py): rows are 8 x u32, `extras` = the load lists ((n_rows/NB + 2) x NLD
                            words), shape = NB | NLD << 8, n_lds = 2*NB + 2*NLD
            bit program     (if n_bit_programs, hip_elements/bitsched.py)  8 x u32: ring, n_vrows, n_slots lo, hi, cache, n_asserts, 2, 0;
                            records n_vrows*64 x 2 u32; command blocks n_vrows/8 x 24 u32; signal -> slot; assertion slots
            emitted 256-bit code (optional, after everything else; hip_elements/fpjit.py): see the end of this function
            emitted code    (if n_bit_programs == 2, hip_elements/bitjit.py)  10 x u32: 2, n_slots lo, hi, code bytes, flags (bit 0:
                            the fused R1CS check covers every constraint), VGPRs, AccVGPRs, bytes of the AUDIT code object
                            (0: none), CRC-32 and byte length of the constraint section of the .r1cs the checks were built from
                            (0, 0: unknown; write_r1cs returns them - cw_load trusts fused checks only when the .r1cs it is given
                            has the same section); signal -> slot; the gfx950 code object (ELF), padded to 4 bytes; then the
                            audit code object (bitjit.lower_jit(audit_of=): the check's gates on loaded rows), padded to 4 bytes
                            (format 1 of earlier releases: 8 words, no audit, no identity)
    """
    if isinstance(tapes, Tape):
        tapes = [tapes]
    t0 = tapes[0]
    n64 = (t0.q.bit_length() + 63) // 64
    for t in tapes[1:]:
        assert t.consts == t0.consts and t.n_signals == t0.n_signals, "variants must come from the same circuit"
        assert t.lconsts == t0.lconsts, "variants must share the limb-form constant table"
        assert bool(getattr(t, "mont", False)) == bool(getattr(t0, "mont", False)), "variants must share the value form"
    with open(path, "wb") as f:
        f.write(TAPE_MAGIC + struct.pack("<III", TAPE_VERSION, n64, len(tapes)))
        f.write(t0.q.to_bytes(8 * n64, "little"))
        f.write(struct.pack("<16I", t0.n_signals, t0.n_witness, len(t0.consts), t0.main_input_start, t0.n_main_inputs,
                            len(t0.inputs), hashmap_size(len(t0.inputs)), t0.rbits | (0x10000 if getattr(t0, "mont", False) else 0),
                            len(t0.lconsts), t0.n_pub_in,
                            (2 if jit is not None else 1) if bittape is not None else 0, len(t0.functions),
                            getattr(t0, "n_dat_consts", 0xFFFFFFFF), getattr(t0, "n_io_templates", 0),
                            len(getattr(t0, "log_prog", ())), sum(1 for _, items in getattr(t0, "log_prog", ()) for it in items if it[0] == "v")))
        f.write(b"".join(c.to_bytes(8 * n64, "little") for c in t0.consts))
        f.write(b"".join(c.to_bytes(8 * n64, "little") for c in t0.lconsts))
        f.write(np.asarray(t0.witness2signal, dtype="<u4").tobytes())
        for name, start, size in t0.inputs:
            b = name.encode()
            f.write(struct.pack("<I", len(b)) + b + struct.pack("<II", start, size))
        for n_regs, fcode, native in t0.functions:       # circom functions with run-time control flow (lower.py D_CALL)
            kind, n_, k_, modulus = native or (0, 0, 0, 0)
            f.write(struct.pack("<5I", n_regs, len(fcode), kind, n_, k_) + int(modulus).to_bytes(32, "little"))
            f.write(np.ascontiguousarray(fcode, dtype="<u4").tobytes())
        for at, items in getattr(t0, "log_prog", ()):
            f.write(struct.pack("<2I", at, len(items)))
            for kind, v in items:
                if kind == "s":
                    b = t0.log_strings[v].encode()
                    f.write(struct.pack("<2I", 0, len(b)) + b)
                else:
                    f.write(struct.pack("<2I", 1, v))
        for t in tapes:
            kind = getattr(t, "kind", 0)
            shape = (t.pipe[0] | (t.pipe[1] << 8)) if kind == 1 else len(t.seqs)
            f.write(struct.pack("<8I", t.n_strands, t.n_tslots, len(t.rows), len(t.extras), t.n_lds, len(t.terms), kind, shape))
            f.write(np.asarray(t.stream_off, dtype="<u4").tobytes())
            f.write(np.asarray(t.extra_off, dtype="<u4").tobytes())
            f.write(np.asarray(t.term_off, dtype="<u4").tobytes())
            f.write(np.ascontiguousarray(t.rows, dtype="<u4").tobytes())
            f.write(np.asarray(t.extras, dtype="<u4").tobytes())
            f.write(np.ascontiguousarray(t.terms, dtype="<u4").tobytes())
            if kind == 0:
                f.write(np.asarray(t.seq_off, dtype="<u4").tobytes())
                f.write(np.asarray(t.seqs, dtype="<u4").tobytes())
        if bittape is not None:
            assert bittape.n_signals == t0.n_signals and not getattr(t0, "mont", False)
            f.write(struct.pack("<8I", bittape.ring, bittape.n_vrows, bittape.n_slots & 0xFFFFFFFF, bittape.n_slots >> 32,
                                bittape.cache, len(bittape.assert_slots), 2, 0))
            f.write(np.ascontiguousarray(bittape.recs, dtype="<u4").tobytes())
            f.write(np.ascontiguousarray(bittape.cmds, dtype="<u4").tobytes())
            f.write(np.ascontiguousarray(bittape.sig_slot, dtype="<u4").tobytes())
            f.write(np.ascontiguousarray(bittape.assert_slots, dtype="<u4").tobytes())
        if jit is not None:
            assert bittape is not None and jit.code is not None and jit.n_signals == t0.n_signals
            audit = getattr(jit, "audit_code", None) or b""
            rid = r1cs_id or (0, 0)
            # the chunk stride is an immediate of both code objects (bitjit.to_asm): it must be the row count this header carries,
            # which every other kernel that walks the table takes as the stride (compiler.emit_jit records what it assembled)
            for what in ("code_stride", "audit_stride"):
                baked = getattr(jit, what, None)
                if baked is not None and (what == "code_stride" or audit):
                    assert baked == jit.n_slots * 256, "emitted %s %d != header rows %d x 256" % (what, baked, jit.n_slots)
            f.write(struct.pack("<10I", 2, jit.n_slots & 0xFFFFFFFF, jit.n_slots >> 32, len(jit.code), 1 if jit.check_complete else 0,
                                jit.n_vgpr, jit.n_agpr, len(audit), rid[0], rid[1] & 0xFFFFFFFF))
            f.write(np.ascontiguousarray(jit.sig_slot, dtype="<u4").tobytes())
            f.write(jit.code + b"\0" * (-len(jit.code) % 4))
            f.write(audit + b"\0" * (-len(audit) % 4))
        if fpjit:
            # emitted 256-bit code of strand variants (hip_elements/fpjit.py), an optional trailing section: "FPJT" | u32 format
            # 1 | u32 n | per program 8 x u32 {n_strands, code bytes, LDS bytes, scratch bytes, VGPRs, bitmap words, 0, 0} + the
            # fused-check bitmap (bit i: the code itself recomputes row i of the .r1cs) + the code object
            assert len(fpjit) <= 8
            f.write(b"FPJT" + struct.pack("<II", 1, len(fpjit)))
            for fp in fpjit:
                assert fp.code is not None and any(t.n_strands == fp.n_strands and getattr(t, "kind", 0) == 0 for t in tapes)
                cov = np.zeros((len(fp.covered) + 31) // 32, dtype="<u4")
                for i, cbit in enumerate(fp.covered):
                    if cbit:
                        cov[i >> 5] |= np.uint32(1 << (i & 31))
                rid = (r1cs_id or (0, 0)) if any(fp.covered or ()) else (0, 0)
                f.write(struct.pack("<8I", fp.n_strands, len(fp.code), fp.lds_bytes, fp.scratch_bytes, getattr(fp, "n_vgpr", 128), len(cov),
                                    rid[0], rid[1] & 0xFFFFFFFF))
                f.write(cov.tobytes())
                f.write(fp.code + b"\0" * (-len(fp.code) % 4))


def _le_key(k: int) -> bytes:
    # r1cs_writer.rs:49-72 orders wire ids by their little-endian byte strings
    return k.to_bytes(max(1, (k.bit_length() + 7) // 8), "little")


def _lc_block(lc: dict, fs: int) -> bytes:
    items = sorted(((k, v) for k, v in lc.items() if v), key=lambda kv: _le_key(kv[0]))
    out = [struct.pack("<I", len(items))]
    for k, v in items:
        out.append(struct.pack("<I", k) + v.to_bytes(fs, "little"))
    return b"".join(out)


def write_r1cs(path, fc: FlatCircuit, wire_of_signal=None):
    """wire_of_signal: optional map signal id -> wire id (identity at --O0)."""
    q = fc.fp.q
    bits = q.bit_length()
    fs = bits // 8 if bits % 64 == 0 else (bits // 64 + 1) * 8      # dag/src/r1cs_porting.rs:7-11
    cons = []
    for a, b, c in fc.constraints:
        if wire_of_signal is not None:
            a = {wire_of_signal[k]: v for k, v in a.items()}
            b = {wire_of_signal[k]: v for k, v in b.items()}
            c = {wire_of_signal[k]: v for k, v in c.items()}
        cons.append(_lc_block(a, fs) + _lc_block(b, fs) + _lc_block(c, fs))
    n_wires = fc.n_signals
    sec2 = b"".join(cons)
    sec1 = struct.pack("<I", fs) + q.to_bytes(fs, "little") + struct.pack(
        "<IIIIQI", n_wires, fc.n_outputs, fc.n_pub_in, fc.n_prv_in, n_wires, len(cons))
    sec3 = np.arange(n_wires, dtype="<u8").tobytes()
    with open(path, "wb") as f:
        f.write(b"r1cs" + struct.pack("<II", 1, 3))
        for typ, body in ((2, sec2), (1, sec1), (3, sec3)):
            f.write(struct.pack("<IQ", typ, len(body)))
            f.write(body)
    import zlib
    return zlib.crc32(sec2) & 0xFFFFFFFF, len(sec2)     # identity of the constraint system (write_tape r1cs_id=)


def write_sym(path, fc: FlatCircuit):
    names = fc.signal_names()
    # component ids of the .sym file: tree post-order (docs formats/sym.md: main.c = 0, main = 1)
    n = len(fc.comp_inst)
    children = [[] for _ in range(n)]
    for ci in range(1, n):
        children[fc.comp_father[ci]].append(ci)
    post = [0] * n
    counter = 0
    stack = [(0, 0)]
    while stack:
        node, k = stack.pop()
        if k < len(children[node]):
            stack.append((node, k + 1))
            stack.append((children[node][k], 0))
        else:
            post[node] = counter
            counter += 1
    comp_of = np.zeros(fc.n_signals, dtype=np.int64)
    insts = fc.prog.inst_list
    for ci in range(n):
        base = fc.comp_sigstart[ci]
        comp_of[base:base + insts[fc.comp_inst[ci]].n_local] = post[ci]
    with open(path, "w") as f:
        for s in range(1, fc.n_signals):
            f.write("%d,%d,%d,%s\n" % (s, s, comp_of[s], names[s]))


def wtns_bytes(q: int, values) -> bytes:
    """values: iterable of canonical ints (witness order)."""
    n8 = 8 * ((q.bit_length() + 63) // 64)
    vals = list(values)
    out = [b"wtns", struct.pack("<II", 2, 2), struct.pack("<IQ", 1, 8 + n8), struct.pack("<I", n8),
           q.to_bytes(n8, "little"), struct.pack("<I", len(vals)), struct.pack("<IQ", 2, n8 * len(vals))]
    out += [v.to_bytes(n8, "little") for v in vals]
    return b"".join(out)
