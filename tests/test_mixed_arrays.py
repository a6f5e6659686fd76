"""SURVEY row a13, `Mapped` locations: component arrays whose elements are instances of ONE template with DIFFERENT
parameters (a `Mixed` cluster).  The reference addresses their signals through the io-map of the `.dat` and runs them
through `_functionTable`; this front-end resolves every access at trace time, so what has to be right is (1) the io-map
section of the `.dat` (the REFERENCE RUNTIME parses it and executes the oracle's emitted C++ through it), (2) the loader's
validation of that section, (3) the witness itself - oracle, reference runtime and GPU byte for byte."""
import struct

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.basic import MixedArray
from circom_amd.hip_elements.writers import wtns_bytes, dat_io_map
from oracle.tape_eval import eval_flat

WIDTHS = ((2, 3), (1, 5), (3, 2), (2, 3))          # elements 0 and 3 are the SAME instance, 1 and 2 differ


def _inputs(q, n, seed):
    rng = np.random.default_rng(seed)
    return [[int.from_bytes(rng.bytes(32), "little") % q for _ in range(8)] for _ in range(n)] + [[0] * 8, [q - 1] * 8, list(range(1, 9))]


def _want(fc, row):
    sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code,
                            {fc.main_input_start + k: v for k, v in enumerate(row)})
    assert failed is None
    return sig


def test_io_map_of_a_mixed_cluster(tmp_path):
    cp = compile_program(Program(MixedArray(WIDTHS)), str(tmp_path), "mixed", sym=False, strands=(1,))
    fc = cp.flat
    # three distinct instances of PowerSums are in the cluster; per instance two io signals: out (2-dimensional), in
    assert len(fc.io_map) == 3
    shapes = sorted((tuple(d[1] for d in defs), tuple(d[0] for d in defs)) for _, defs in fc.io_map)
    assert shapes == [(((1, 5), (1,)), (0, 5)), (((2, 3), (2,)), (0, 6)), (((3, 2), (3,)), (0, 6))]
    main = fc.prog.main
    assert main.mixed_children == {0, 1, 2, 3}
    # the section as the reference lays it out: ids, then per template {n, per signal: offset, dims - 1, lengths[1:], size, bus}
    blob = dat_io_map(fc.io_map)
    words = struct.unpack("<%dI" % (len(blob) // 4), blob)
    assert list(words[:3]) == [tid for tid, _ in fc.io_map]
    assert words[3] == 2 and words[4] == 0 and words[5] == 1          # first template: 2 signals; out at 0 with ONE extra length
    dat = open(cp.dat_path, "rb").read()
    assert dat.endswith(blob)
    # the witness: sum over elements of (i + 2) * in^m
    q = fc.fp.q
    row = list(range(1, 9))
    sig = _want(fc, row)
    k0, exp = 0, 0
    for i, (n, m) in enumerate(WIDTHS):
        for j in range(n):
            exp += (i + 2) * pow(row[k0 + j], m, q)
        k0 += n
    assert sig[1] == exp % q
    # the loader reads and validates the section
    from circom_amd import runtime as rt
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    L = rt.lib()
    assert L.cw_io_map_size(c.h) == 3
    for tid, defs in fc.io_map:
        for code, d in enumerate(defs):
            assert L.cw_io_map_offset(c.h, tid, code) == d[0]
        assert L.cw_io_map_offset(c.h, tid, len(defs)) == -1
    c.close()
    for cut in (4, 8, 12):                                         # truncated / padded sections are rejected
        (tmp_path / "bad.dat").write_bytes(dat[:-cut])
        with pytest.raises(rt.CwError):
            rt.Circuit(cp.tape_path, tmp_path / "bad.dat", cp.r1cs_path)
    (tmp_path / "bad.dat").write_bytes(dat + b"\0\0\0\0")
    with pytest.raises(rt.CwError):
        rt.Circuit(cp.tape_path, tmp_path / "bad.dat", cp.r1cs_path)
    bad = bytearray(dat)
    at = len(dat) - len(blob) + 4 * 4                               # offset of the first signal of the first template
    bad[at:at + 4] = struct.pack("<I", fc.n_signals)
    (tmp_path / "bad.dat").write_bytes(bytes(bad))
    with pytest.raises(rt.CwError):
        rt.Circuit(cp.tape_path, tmp_path / "bad.dat", cp.r1cs_path)


def test_reference_runtime_executes_the_cluster_through_the_io_map(tmp_path, ref_dir_bn128):
    from oracle import ref_build
    cp = compile_program(Program(MixedArray(WIDTHS)), str(tmp_path), "mixed", sym=False, strands=(1,))
    try:
        ref_build.build_circuit(cp)
    except RuntimeError as e:
        pytest.skip(str(e))
    src = (ref_build.ref_dir(cp.flat.prime) / "mixed.cpp").read_text() if hasattr(ref_build, "ref_dir") else ""
    if src:
        assert "templateInsId2IOSignalInfo" in src and "(*_functionTable[" in src and "get_size_of_io_map() {return 3;}" in src
    fc = cp.flat
    q = fc.fp.q
    ins = _inputs(q, 6, 5)
    raw = b"".join(v.to_bytes(32, "little") for row in ins for v in row)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(cp, raw, len(ins), 1, wtns_prefix=pre)
    for i, row in enumerate(ins):
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(q, _want(fc, row)), i
    # the process-level CLI too (it loads the .dat the way users' binaries do)
    out = tmp_path / "cli.wtns"
    r = ref_build.run_cli(cp, '{"x": [%s]}' % ",".join('"%d"' % v for v in ins[0]), out)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == wtns_bytes(q, _want(fc, ins[0]))


@pytest.mark.gpu
def test_gpu_mixed_cluster_matches_oracle(tmp_path):
    from circom_amd import runtime as rt
    cp = compile_program(Program(MixedArray(WIDTHS)), str(tmp_path), "mixed", sym=False, strands=(1, 4))
    fc = cp.flat
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    ins = _inputs(fc.fp.q, 70, 9)
    b = c.batch(len(ins))
    b.set_inputs(ins)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i in (0, 1, 64, len(ins) - 1):
        assert b.witness(i) == _want(fc, ins[i])
        b.write_wtns(i, tmp_path / "g.wtns")
        assert (tmp_path / "g.wtns").read_bytes() == wtns_bytes(fc.fp.q, _want(fc, ins[i]))
    b.close(); c.close()
