"""`python -m circom_amd.hip_backend <name>.cwf [-o DIR]` - the hip_elements back-end as a process: lowers a flat circuit
(`.cwf`, circom_amd/cwf.py: what a front-end's `--hip` target hands over) into the files the runtime loads:
`<name>.cwt` (schedule variants + bit-plane program + emitted code), `<name>.dat` (reference layout), `<name>.r1cs`.

This is the seam the Rust producer calls (integration/code_producers/src/hip_elements/mod.rs `HipProducer::finish`): the
role `generic/makefile` + `g++` play for the `--c` target (compilation_user.rs:34-99 writes C++ and leaves the compile to
make; `--hip` writes the .cwf and runs this)."""
from __future__ import annotations

import argparse
import os
import sys


def lower_cwf(cwf_path: str, outdir: str, name: str, strands=(1, 4, 16), bits="auto"):
    from . import compiler
    from .cwf import read_cwf
    from .hip_elements import writers
    from .hip_elements.lower import lower
    fc = read_cwf(cwf_path)
    os.makedirs(outdir, exist_ok=True)
    bittape, net = (None, None) if os.environ.get("CW_BITS", "1") == "0" else compiler.lower_bitplane_net(fc, bits)
    jp = compiler.emit_jit(net, fc) if bittape is not None else None      # the gate network as emitted code
    del net
    mont = False if bittape is not None else compiler.choose_mont(fc)
    if bittape is not None and fc.n_signals >= compiler.BITS_KEEP_STRANDS_BELOW:
        strands = (1,)
    tapes = [lower(fc, n_strands=s, mont=mont) for s in strands]
    p = lambda ext: os.path.join(outdir, name + ext)
    fps = compiler.emit_fpjit(tapes, fc, False if bittape is not None else "auto")             # the rows as emitted code
    rid = writers.write_r1cs(p(".r1cs"), fc)
    writers.write_tape(p(".cwt"), tapes, bittape, jp, fps, r1cs_id=rid)
    writers.write_dat(p(".dat"), fc)
    return p(".cwt"), p(".dat"), p(".r1cs")


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m circom_amd.hip_backend")
    ap.add_argument("cwf")
    ap.add_argument("-o", "--outdir", default=None)
    ap.add_argument("--strands", default="1,4,16")
    args = ap.parse_args(argv)
    name = os.path.splitext(os.path.basename(args.cwf))[0]
    outdir = args.outdir or os.path.dirname(os.path.abspath(args.cwf))
    for f in lower_cwf(args.cwf, outdir, name, tuple(int(x) for x in args.strands.split(","))):
        print(f)
    return 0


if __name__ == "__main__":
    sys.exit(main())
