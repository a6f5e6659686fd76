"""`python -m circom_amd.circom <file>.circom [options]` - the compiler driver for circom SOURCE files, with the option names
of the reference's command line (circom/src/input_user.rs:218-420; driver order circom/src/main.rs:18-60: parse ->
type analysis -> execution -> compilation):

    --r1cs  --sym  --json   outputs of the constraint system          (constraint_writers: r1cs_writer.rs, sym_writer.rs,
                                                                         json_writer.rs)
    --hip                   the MI355X witness calculator: <name>_hip/<name>.{cwt,dat} (+ .r1cs for its check kernel) - the
                            target that stands where --c / --wasm stand (compilation_user.rs:34-99)
    -o DIR                  output directory (default .)
    -l DIR                  library directory for `include` (repeatable; include_logic.rs)
    -p / --prime NAME       bn128 (default) bls12381 goldilocks grumpkin pallas vesta secq256r1 bls12377
    --O1                    (the DEFAULT, as in the reference: input_user.rs:264-283) constant and renaming simplifications:
                            `.r1cs` / `.sym` / JSON are written from the simplified system (frontend/circom_simplify.py), the
                            witness keeps fewer signals; <name>_hip/<name>.w2s lists them (u32 LE) - the device still generates
                            and checks the full system, `reduce_wtns` cuts the O1 `.wtns` out of its output
    --O0                    no simplification: every signal is a witness entry and every constraint is kept
    --O2                    refused (the reference's Gaussian elimination with its signal-choice heuristics is not implemented)
    --inspect               the warnings of dag/src/constraint_correctness_analysis.rs (warning[CA01] local signals, warning[CA02]
                            inputs / outputs of sub-components that appear in no constraint of the template), then one line per
                            template instance; the summary lines (template instances, constraints, inputs / outputs / wires /
                            labels: dag/src/lib.rs:417-456) are always printed, as by the reference

The reference's own front-end is Rust (parser, type_analysis, constraint_generation, dag, compiler); this image has no
cargo, so the language is implemented again here (frontend/circom_lang.py, circom_exec.py, circom_rt.py) on top of the
tracing front-end the Python circuits already use.  Exit status: 0, or 1 after printing `error: ...` (the reference
prints a Report and "previous errors were found").
"""
from __future__ import annotations

import argparse
import os
import sys


def compile_file(path, outdir=".", libs=(), prime="bn128", r1cs=False, sym=False, json_out=False, hip=False, inspect=False,
                 strands=None, out=None, level="O1"):
    from .frontend.circom_exec import program_from_file
    out = out or sys.stdout
    from .frontend.flatten import flatten
    from .hip_elements import writers
    name = os.path.splitext(os.path.basename(path))[0]
    prog = program_from_file(path, libs, prime, inspect=inspect)
    fc = flatten(prog)
    for w in prog.world.warnings:                    # --inspect: dag/src/constraint_correctness_analysis.rs (CA01 / CA02)
        print("warning[%s]: %s" % ("CA02" if "ubcomponent" in w else "CA01", w), file=out)
    for w in prog.world.typing_warnings:             # arrays of different lengths into variables (execute.rs:3949-3965)
        print("warning[T3001]: %s" % w, file=out)
    os.makedirs(outdir, exist_ok=True)
    written = []
    # the summary lines of the reference (circom/src/execution_user.rs + dag/src/lib.rs:417-456)
    sm = None
    if level == "O1":
        from .frontend import circom_simplify
        sm = circom_simplify.simplify_o1(fc)
    cons = fc.constraints if sm is None else sm.constraints
    n_lin = sum(1 for a, b, c in cons if not a or not b)
    print("template instances: %d" % len(prog.inst_list), file=out)
    print("non-linear constraints: %d" % (len(cons) - n_lin), file=out)
    print("linear constraints: %d" % n_lin, file=out)
    print("public inputs: %d" % fc.n_pub_in, file=out)
    n_prv = fc.n_prv_in if sm is None else sm.n_prv_in
    print("private inputs: %d%s" % (fc.n_prv_in, "" if n_prv == fc.n_prv_in else " (%d belong to witness)" % n_prv), file=out)
    print("public outputs: %d" % fc.n_outputs, file=out)
    print("wires: %d" % (fc.n_signals if sm is None else sm.n_wires), file=out)
    print("labels: %d" % fc.n_signals, file=out)
    for line in getattr(prog, "world").compile_log:
        print(line, file=out)
    if r1cs:
        p = os.path.join(outdir, name + ".r1cs")
        if sm is None:
            writers.write_r1cs(p, fc)
        else:
            circom_simplify.write_r1cs(p, sm)
        written.append(p)
    if sym:
        p = os.path.join(outdir, name + ".sym")
        if sm is None:
            writers.write_sym(p, fc)
        else:
            circom_simplify.write_sym(p, sm)
        written.append(p)
    if json_out:
        # constraint_writers/src/json_writer.rs: {"constraints": [[A, B, C], ...]} with decimal strings
        p = os.path.join(outdir, name + "_constraints.json")
        from .frontend.circom_simplify import constraints_json
        with open(p, "w") as fh:
            fh.write(constraints_json(cons))
        written.append(p)
    if hip:
        from . import compiler
        hdir = os.path.join(outdir, name + "_hip")
        kw = {} if strands is None else {"strands": strands}
        cp = compiler.compile_program(prog, hdir, name, sym=False, **kw)
        written += [cp.tape_path, cp.dat_path, cp.r1cs_path]
        fc.compiled = cp
        if sm is not None:
            # the device generates and checks the full system; the simplified witness is its output read through this list
            import numpy as np
            p = os.path.join(hdir, name + ".w2s")
            np.asarray(sm.witness2signal, dtype="<u4").tofile(p)
            written.append(p)
    fc.simplified = sm
    for p in written:
        print("Written successfully: %s" % p, file=out)
    if inspect:
        for inst in prog.inst_list:
            print("  %s(%s): %d signals, %d constraints" % (inst.name, ", ".join(_short(v) for v in inst.params),
                                                           inst.n_total, len(inst.constraints)), file=out)
    print("Everything went okay", file=out)
    return fc, written


def _short(v):
    s = str(v)
    return s if len(s) <= 24 else s[:21] + "..."


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m circom_amd.circom")
    ap.add_argument("input")
    ap.add_argument("--r1cs", action="store_true")
    ap.add_argument("--sym", action="store_true")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--hip", action="store_true")
    ap.add_argument("--c", action="store_true", help="refused: this tree's target is --hip")
    ap.add_argument("--wasm", action="store_true", help="refused: this tree's target is --hip")
    ap.add_argument("-o", "--output", default=".")
    ap.add_argument("-l", action="append", default=[], dest="libs")
    ap.add_argument("-p", "--prime", default="bn128")
    ap.add_argument("--O0", action="store_true")
    ap.add_argument("--O1", action="store_true")
    ap.add_argument("--O2", action="store_true")
    ap.add_argument("--inspect", action="store_true")
    ap.add_argument("--strands", default=None)
    args = ap.parse_args(argv)
    from .frontend.circom_lang import CircomSyntaxError
    from .frontend.dsl import CircuitError
    if args.c or args.wasm:
        print("error: --c / --wasm are the reference's targets; this front-end produces --hip", file=sys.stderr)
        return 1
    if args.O2:
        print("error: --O2 is not implemented (use --O1, the default, or --O0)", file=sys.stderr)
        return 1
    try:
        compile_file(args.input, args.output, args.libs, args.prime, args.r1cs, args.sym, args.json, args.hip, args.inspect,
                     None if args.strands is None else tuple(int(x) for x in args.strands.split(",")),
                     level="O0" if args.O0 else "O1")
    except (CircomSyntaxError, CircuitError, FileNotFoundError) as ex:
        print("error: %s" % ex, file=sys.stderr)
        print("previous errors were found", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
