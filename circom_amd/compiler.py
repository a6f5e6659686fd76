"""`circom <file> --r1cs --sym --hip` stand-in: trace a Program and emit every artefact the new back-end
produces (role of circom/src/compilation_user.rs:31-139 + execution_user.rs:47-55 for the --hip target)."""
from __future__ import annotations

import os
from dataclasses import dataclass

from . import opcodes as O
from .frontend.dsl import Program
from .frontend.flatten import flatten, FlatCircuit
from .hip_elements.lower import lower, Tape
from .hip_elements import writers


@dataclass
class Compiled:
    name: str
    dir: str
    tape_path: str
    dat_path: str
    r1cs_path: str
    sym_path: str
    flat: FlatCircuit
    tape: Tape
    bittape: object = None          # hip_elements.bitsched.BitTape when the circuit got a bit-plane program
    jit: object = None              # hip_elements.bitjit.JitProgram: the same network as emitted gfx950 code (large batches)
    fpjit: tuple = ()               # hip_elements.fpjit.FpJitProgram per strand variant: the rows as emitted gfx950 code


DEFAULT_STRANDS = (1, 4, 16)


def strands_for(batch: int):
    """The one variant cw_batch_create picks for `batch` instances (csrc/cw_host.cpp: most strands with
    groups * S <= 8192) — lowering a 1M-constraint circuit takes about a minute per variant, so callers that know
    their batch lower only this one."""
    groups = (batch + 63) // 64
    for s in (16, 4):
        if groups * s <= 8192:
            return (s,)
    return (1,)


def parallel_speedup(tape) -> float:
    """work / critical path of a strand schedule under the measured costs of interpreted rows (lower.py _COST_INTERP; the twin of
    cw_host.cpp's choice in cw_batch_create): per barrier epoch the busiest strand's cost, summed, against the cost of all
    rows.  A circuit that is one long call plus a few rows (BigMultModP) gets ~1 whatever the strand count: its single-strand
    variant (which has an emitted form) should be in the tape as well."""
    import numpy as np
    from .hip_elements import lower as L
    rows, so, S = np.asarray(tape.rows), tape.stream_off, tape.n_strands
    if S <= 1 or len(rows) == 0:
        return 1.0
    fixed = {L.D_MULC: 8.0, L.D_MADDC: 8.0, L.D_MADD: 8.0, L.D_MUL2: 10.0, L.D_IDIV: 23.0, L.D_MOD: 23.0, L.D_INV: 180.0, L.D_POW: 6000.0}
    per = []
    n_ep = 0
    for s_ in range(S):
        r = rows[int(so[s_]):int(so[s_ + 1])]
        op = r[:, 0] & 0xFF
        nx = (r[:, 0] >> L.SH_NX) & 0xFFF
        c = np.full(len(r), 6.0)
        for k, v in fixed.items():
            c[op == k] = v
        c[op == L.D_LINSUM] = 6.0 + 4.8 * r[op == L.D_LINSUM, 2]
        c[op == L.D_DOTC] = 6.0 + 6.0 * r[op == L.D_DOTC, 2]
        c[op == L.D_BITS] = 6.0 + 0.3 * nx[op == L.D_BITS]
        for i in np.nonzero(op == L.D_CALL)[0]:
            nat = tape.functions[int(r[i, 2])][2]
            c[i] = 20000.0 if nat is None else 130.0 if nat[0] == 4 else 170.0
        c[op == L.D_BARRIER] = 0.0
        ep = np.cumsum(op == L.D_BARRIER)
        e = np.bincount(ep, weights=c)
        per.append(e)
        n_ep = max(n_ep, len(e))
    E = np.zeros((S, n_ep))
    for s_, e in enumerate(per):
        E[s_, :len(e)] = e
    crit = float(E.max(axis=0).sum())
    return float(E.sum()) / crit if crit > 0 else 1.0


def choose_mont(fc: FlatCircuit) -> bool:
    """Montgomery-form signals (lower.py pass A6) pay off for arithmetic circuits: every product of two run-time values
    saves one of its two Montgomery products, every value that an integer operator touches (bit extraction, shifts, bitwise,
    ordering comparisons, integer division) costs one conversion, and the short paths for small values (bit-level circuits:
    products and sums of 0/1 signals) are lost.  Chosen when the run-time products outnumber the integer operators 4 : 1."""
    code = fc.code
    op = code["op"]
    if (op == O.CALL).any():
        return False
    n_mul = int(((op == O.MUL) & (code["ak"] != O.K_CONST) & (code["bk"] != O.K_CONST)).sum()) + int((op == O.DIV).sum())
    int_ops = (O.IDIV, O.MOD, O.POW, O.SHL, O.SHR, O.BAND, O.BOR, O.BXOR, O.BNOT, O.LT, O.GT, O.LEQ, O.GEQ)
    n_int = sum(int((op == o).sum()) for o in int_ops)
    return n_mul > 0 and n_mul >= 4 * n_int


BITS_AUTO_MIN_SIGNALS = 4096
BITS_AUTO_MIN_SIGNALS_PER_GATE = 16     # "auto" wants at least one gate per 16 signals (see lower_bitplane)
BITS_KEEP_STRANDS_BELOW = 200_000       # signals: smaller circuits keep every strand variant next to the bit program


def lower_bitplane_net(fc: FlatCircuit, bits="auto"):
    """(bit-plane program | None, its gate network | None) of a circuit whose signals are all boolean for 0/1 inputs (SHA-256 and
    friends).  bits: True = whenever the analysis succeeds, False = never, "auto" = only for circuits large enough to matter
    (an instance whose inputs are not 0/1 is re-run by the 256-bit schedule, so tiny arithmetic circuits gain nothing).
    The network goes to the code emitter (emit_jit)."""
    if bits is False or (bits == "auto" and fc.n_signals < BITS_AUTO_MIN_SIGNALS) or fc.n_main_inputs == 0:
        return None, None
    if (fc.code["op"] == O.LOG).any():
        return None, None   # logged values live in the 256-bit table (hidden signals); the bit table has no place for them
    from .hip_elements.bitblast import bitblast
    from .hip_elements.bitsched import lower_bits
    net = bitblast(fc)
    if net is None:
        return None, None
    # Evidence that the circuit really is bit-level, tested BEFORE the mapping and the two scheduling passes (an
    # arithmetic circuit with range checks used to pay the whole lowering just to be rejected)
    if bits == "auto" and net.stats["gates"] * BITS_AUTO_MIN_SIGNALS_PER_GATE < fc.n_signals:
        return None, None
    from .hip_elements.bitmap import map_network
    bt = lower_bits(map_network(net), fc)
    # Evidence that the circuit really is bit-level: the analysis only proves "boolean IF the inputs are 0/1" - a Num2Bits or
    # range check on field-valued inputs passes it with a handful of gates (everything else is discharged symbolically) and
    # would then send every instance through the non-boolean fallback.  A bit-level circuit has gates in proportion to its
    # signals (SHA-256: 0.7 per signal).
    if bt is not None and bits == "auto" and bt.stats["gates"] * BITS_AUTO_MIN_SIGNALS_PER_GATE < fc.n_signals:
        return None, None
    return bt, (net if bt is not None else None)


def lower_bitplane(fc: FlatCircuit, bits="auto"):
    """the bit-plane program alone (see lower_bitplane_net)"""
    return lower_bitplane_net(fc, bits)[0]


def _emit_failure(what, ex, strict):
    """an emitter that cannot produce its code object (no ROCm LLVM on this host, an assembler error): `auto` keeps the tape
    valid without emitted code - the interpreting kernels run every schedule - and says so; an explicit request re-raises"""
    if strict:
        raise ex
    import warnings
    warnings.warn("circom_amd: %s not emitted (%s: %s); the tape carries the interpreted program only" % (what, type(ex).__name__, str(ex)[:200]))

JIT_AUTO_MIN_GATES = 20_000         # "auto": circuits below this never see batches where the emitted code wins
JIT_LOOP_MIN_GATES = 1_500_000      # "auto": networks above this get ONE looped body per repeated template (see emit_jit)


def emit_jit(net, fc, jit="auto"):
    """The bit-plane program as emitted gfx950 code (hip_elements/bitjit.py), assembled; None when not wanted / nothing to
    evaluate.  jit: True, False or "auto" (circuits with at least JIT_AUTO_MIN_GATES gates); CW_JIT=0/1 overrides."""
    if os.environ.get("CW_JIT"):
        jit = os.environ["CW_JIT"] != "0"
    if net is None or jit is False or (jit == "auto" and net.stats.get("gates", 0) < JIT_AUTO_MIN_GATES):
        return None
    from .hip_elements import bitjit
    import subprocess
    # ONE body per repeated template (the reference's code structure: template.rs:160-474): when the circuit instantiates a large
    # template several times (the compression blocks of SHA-256), the emitter works on a network in which those instances' input
    # signals are PORTS - every instance then has the same gates - and emits the body once, in a loop (bitjit.lower_jit).  The
    # interpreter's program keeps the fully folded network it was given.  CW_JIT_LOOP=0: straight-line code as before.
    # Measured (profiles/r06a_*): the looped code of the 5-block SHA-256 is 2.4 MB instead of 13 MB but 10 % SLOWER
    # (12.4 -> 13.6 ms at 2^21 instances): the kernel is bound by the rows it stores, ports keep the IV / the padding from folding
    # into the first and last block (+9 % rows), and straight-line code is fetched as fast as a 2 MB body (tools/ubench_loop.hip).
    # So "auto" rolls only what would not be practical as a straight line: networks above JIT_LOOP_MIN_GATES gates (the 53-block
    # SHA-256: 137 MB of code unrolled, 2.4 MB + tables rolled).  CW_JIT_LOOP=1 / 0 force it on / off.
    loop_mode = os.environ.get("CW_JIT_LOOP", "auto")
    if loop_mode != "0" and (loop_mode == "1" or net.stats.get("gates", 0) > JIT_LOOP_MIN_GATES):
        ports, marks = bitjit.instance_ports(fc)
        if ports:
            from .hip_elements.bitblast import bitblast
            net_p = bitblast(fc, ports=ports, marks=marks)
            if net_p is not None:
                net = net_p
    jp = bitjit.lower_jit(net, fc)
    if jp is None:
        return None
    # the stand-alone audit of the table this program writes (cw_check_r1cs under CW_R1CS_AUDIT=1 / after cw_device_bits): the
    # check's gates alone on LOADED rows - a second, much smaller code object; its scratch rows extend the chunk.  It is lowered
    # BEFORE either program is printed: the chunk stride (rows per chunk x 256 bytes) is an immediate of the code, and both
    # programs, the tape header and every kernel that walks the table must agree on it (the main program used to be
    # assembled with its own row count and the header then raised to the audit's)
    ja = None
    if os.environ.get("CW_JIT_AUDIT", "1") != "0":
        akw = {}
        if os.environ.get("CW_JIT_AUDIT_REGS"):     # "vgprs,accvgprs" for the audit program: tests starve it so that it spills scratch rows
            nv_, na_ = (int(x) for x in os.environ["CW_JIT_AUDIT_REGS"].split(","))
            akw = {"n_vgpr": nv_, "n_agpr": na_}
        try:
            ja = bitjit.lower_jit(net, fc, audit_of=jp, **akw)
        except bitjit._AuditTooBig:
            # (the audit's iteration wanted more scratch rows than a loop iteration of the main program has to spare)
            ja = bitjit.lower_jit(net, fc, audit_of=jp, loop=False, **akw)
        if ja is not None:
            jp.n_slots = ja.n_slots = max(jp.n_slots, ja.n_slots)
    try:
        jp.code = bitjit.assemble(bitjit.to_asm(jp))
    except (RuntimeError, OSError, subprocess.CalledProcessError) as ex:
        _emit_failure("the bit-plane program's code", ex, jit is True)
        return None
    jp.code_stride = jp.n_slots * bitjit.ROW_BYTES
    if ja is not None:
        try:
            jp.audit_code = bitjit.assemble(bitjit.to_asm(ja))
        except (RuntimeError, OSError, subprocess.CalledProcessError) as ex:
            _emit_failure("the audit program's code", ex, False)
            return jp                           # (the main code keeps the raised stride: the header carries jp.n_slots as well)
        jp.audit_stride = ja.n_slots * bitjit.ROW_BYTES
        jp.stats["audit_instructions"] = ja.stats["instructions"]
        jp.stats["audit_loads"] = ja.stats["prefetched"] + ja.stats["late_loads"]
    return jp


FPJIT_MAX_ROWS = 400_000            # "auto": beyond this the code object (about 100 bytes per row) is not worth its size ...
FPJIT_MAX_ROWS_CALLS = 4_000_000    # ... except for strand schedules around run-time function calls (config 5's verifier: 1.25 M rows):
                                    # the interpreting kernel pays ~5 K clocks per row whatever it computes, the emitted rows a
                                    # fraction of it; such programs are emitted WITHOUT the fused-check twin (twice the code)


def emit_fpjit(tapes, fc, fpjit="auto", fuse_check=True):
    """The strand variants of the 256-bit schedule as emitted gfx950 code (hip_elements/fpjit.py), assembled.  fpjit: True,
    False or "auto" (arithmetic circuits up to FPJIT_MAX_ROWS rows; circuits whose every instance normally takes the
    bit-plane program keep the interpreter for the rare fallback instance); CW_FPJIT=0/1 overrides.  fuse_check: the code
    also recomputes the R1CS rows of the classes it knows right behind the rows that produce their wires (CW_FPJIT_CHECK=0:
    evaluation only, the stand-alone kernel checks every row)."""
    if os.environ.get("CW_FPJIT_CHECK"):
        fuse_check = os.environ["CW_FPJIT_CHECK"] != "0"
    if os.environ.get("CW_FPJIT"):
        fpjit = os.environ["CW_FPJIT"] != "0"
    if fpjit is False:
        return ()
    from .hip_elements import fpjit as FJ
    import subprocess
    out = []
    skipped = []
    for t in tapes:
        if getattr(t, "kind", 0) != 0:
            continue
        big_calls = bool(t.functions) and t.n_strands > 1 and len(t.rows) > FPJIT_MAX_ROWS
        if fpjit == "auto" and len(t.rows) > (FPJIT_MAX_ROWS_CALLS if big_calls else FPJIT_MAX_ROWS):
            continue
        # two programs per variant: the rows alone, and the rows with the R1CS check fused in.  Which one a batch runs is the
        # runtime's choice (cw_batch_create): recomputing the constraints inside the evaluation costs about as many instructions
        # as the evaluation itself - it pays where the launch is throughput-bound (the wires are in registers and caches instead
        # of a second pass over HBM), not where a small batch waits on its dependency chain
        for cons in ((None, fc.constraints) if (fuse_check and fc.constraints and not big_calls) else (None,)):
            spool_dir = None
            try:
                # (beyond ~300 K rows the text goes to a file as it is produced: tens of millions of lines)
                spool = None
                if len(t.rows) > 300_000:
                    import tempfile
                    spool_dir = tempfile.mkdtemp(prefix="cw_fpjit_")
                    spool = os.path.join(spool_dir, "k.s")
                p = FJ.emit(t, constraints=cons, spool_path=spool)
                FJ.assemble(p)
            except NotImplementedError as ex:
                # a variant that is interpreter-only by design (several strands around run-time function calls, the native
                # long_div: fpjit.emit says which): skipped - with a warning under an explicit request, which only fails when
                # NO variant of the circuit could be emitted
                skipped.append(str(ex))
                if fpjit is True:
                    import warnings
                    warnings.warn("circom_amd: the %d-strand variant runs on the interpreting kernel (%s)" % (t.n_strands, ex))
                break
            except (RuntimeError, OSError, subprocess.CalledProcessError) as ex:
                _emit_failure("the rows' code (%d strands)" % t.n_strands, ex, fpjit is True)
                break
            finally:
                if spool_dir is not None:
                    import shutil
                    shutil.rmtree(spool_dir, ignore_errors=True)
            out.append(p)
    if fpjit is True and not out and skipped:
        raise NotImplementedError("no variant of this circuit has an emitted form: " + "; ".join(sorted(set(skipped))))
    return tuple(out)


# The pipelined single-wave variant (hip_elements/pipe.py) is opt-in: measured on MI355X it matches the plain single-strand
# schedule at high occupancy and loses to the multi-strand variants at small batches, where one wave per 64 instances
# leaves most SIMDs idle (Poseidon(2) x 8 192: 1.2 ms against 0.67 ms with 4 strands; NOTES.md round 2).
DEFAULT_PIPE = None


def compile_program(prog: Program, outdir: str, name: str, sym: bool = True, strands=DEFAULT_STRANDS, bits="auto",
                    pipe=DEFAULT_PIPE, mont="auto", jit="auto", fpjit="auto") -> Compiled:
    """strands: strand counts to lower the schedule for (one variant each; the runtime picks per batch).
    pipe: (rows, loads) per batch of the pipelined variant, e.g. (8, 8), which is added last; None = no pipelined variant.
    mont: signals in Montgomery form on the device (True / False / "auto" = choose_mont); CW_MONT=0/1 overrides."""
    os.makedirs(outdir, exist_ok=True)
    fc = flatten(prog)
    if fc.fp.n64 == 1:
        # --prime goldilocks: the reference's 64-bit runtime (goldilocks/fr.hpp, common64/): its own small engine on the device
        # (csrc/cw64.hip), the flat program as it stands (hip_elements/lower64.py), 8-byte file formats
        from .hip_elements import lower64 as L64
        t64 = L64.lower64(fc)
        p64 = lambda ext: os.path.join(outdir, name + ext)
        L64.write_tape64(p64(".cwt"), fc, t64)
        L64.write_dat64(p64(".dat"), fc)
        writers.write_r1cs(p64(".r1cs"), fc)
        if sym:
            writers.write_sym(p64(".sym"), fc)
        return Compiled(name, outdir, p64(".cwt"), p64(".dat"), p64(".r1cs"), p64(".sym"), fc, t64)
    bittape, net = (None, None) if os.environ.get("CW_BITS", "1") == "0" else lower_bitplane_net(fc, bits)   # CW_BITS=0: no bit program in the tape
    jp = emit_jit(net, fc, jit) if bittape is not None else None
    del net
    if bittape is not None:
        # the 256-bit schedule serves the instances re-run with non-boolean inputs - possibly the whole batch (a caller that
        # feeds field-valued inputs): small circuits keep their multi-strand variants, for a 1M-signal circuit one
        # variant is a minute of lowering
        if fc.n_signals >= BITS_KEEP_STRANDS_BELOW:
            strands = (1,)
        pipe = None
    if getattr(fc, "functions", None) and (fc.code["op"] == O.CALL).any():
        pipe = None                         # run-time control flow: program order on the value table (tier 2)
    if os.environ.get("CW_PIPE_SHAPE"):
        pipe = tuple(int(x) for x in os.environ["CW_PIPE_SHAPE"].split(","))
    if bittape is not None:
        mont = False                        # bit-level circuit: the short paths for small values need canonical values
    elif os.environ.get("CW_MONT"):
        mont = os.environ["CW_MONT"] != "0" and not (fc.code["op"] == O.CALL).any()
    elif mont == "auto":
        mont = choose_mont(fc)
    tapes = [lower(fc, n_strands=s, mont=mont) for s in strands]
    if pipe is not None:
        tapes.append(lower(fc, pipe=pipe, mont=mont))
    tape = tapes[0]
    p = lambda ext: os.path.join(outdir, name + ext)
    fps = emit_fpjit(tapes, fc, False if (bittape is not None and fpjit == "auto") else fpjit)
    rid = writers.write_r1cs(p(".r1cs"), fc)          # first: the tape records which constraint system its fused checks belong to
    writers.write_tape(p(".cwt"), tapes, bittape, jp, fps, r1cs_id=rid)
    writers.write_dat(p(".dat"), fc)
    if sym:
        writers.write_sym(p(".sym"), fc)
    return Compiled(name, outdir, p(".cwt"), p(".dat"), p(".r1cs"), p(".sym"), fc, tape, bittape, jp, fps)
