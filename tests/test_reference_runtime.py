"""The strongest pin of the oracle: the reference's own C++ runtime (main.cpp + calcwit.cpp + rendered
generic/fr.cpp, compiled from /root/reference by oracle/Makefile) runs the restated circuits and its
.wtns output is compared byte for byte with the Python oracle (CPU test) and with the HIP path (GPU test)."""
import os

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.basic import Multiplier2, Num2Bits, IsZero
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.hip_elements.writers import wtns_bytes
from oracle import ref_build
from oracle.tape_eval import eval_flat


def _build(tmp_path, prog, name):
    cp = compile_program(prog, str(tmp_path), name, sym=False)
    try:
        ref_build.build_circuit(cp)
    except RuntimeError as e:
        pytest.skip(str(e))
    return cp


def _rand_inputs(q, n, n_in, seed):
    rng = np.random.default_rng(seed)
    return [[int.from_bytes(rng.bytes(32), "little") % q for _ in range(n_in)] for _ in range(n)]


def test_reference_cli_multiplier2_docs_vector(tmp_path, ref_dir_bn128):
    cp = _build(tmp_path, Program(Multiplier2()), "multiplier2")
    out = tmp_path / "w.wtns"
    r = ref_build.run_cli(cp, '{"a": "3", "b": "11"}', out)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == wtns_bytes(cp.flat.fp.q, [1, 33, 3, 11])


def test_reference_runtime_equals_python_oracle_poseidon(tmp_path, ref_dir_bn128):
    cp = _build(tmp_path, Program(Poseidon(2)), "poseidon2")
    fc = cp.flat
    q = fc.fp.q
    ins = _rand_inputs(q, 24, 2, 1) + [[1, 2], [0, 0], [q - 1, q - 1], [5, 2 ** 31 - 1], [2 ** 31, q - 2 ** 31]]
    raw = b"".join(v.to_bytes(32, "little") for row in ins for v in row)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(cp, raw, len(ins), 1, wtns_prefix=pre, stride=1)
    for i, row in enumerate(ins):
        want, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {2: row[0], 3: row[1]})
        assert failed is None
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(q, want), i


def test_reference_runtime_small_circuits_with_bit_ops(tmp_path, ref_dir_bn128):
    # Num2Bits / IsZero exercise the short-int, shift, band, div and select paths of the reference library
    for prog, name, vals in ((Program(Num2Bits(16)), "num2bits16", [0, 1, 2, 255, 65535, 43690]),
                             (Program(IsZero()), "iszero", [0, 1, 2, 12345678901234567890123])):
        cp = _build(tmp_path, prog, name)
        fc = cp.flat
        q = fc.fp.q
        raw = b"".join(v.to_bytes(32, "little") for v in vals)
        pre = str(tmp_path / (name + "_"))
        ref_build.run_loop(cp, raw, len(vals), 1, wtns_prefix=pre)
        for i, v in enumerate(vals):
            want, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start: v})
            assert failed is None
            assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(q, want), (name, v)
    # a failing `===` aborts the reference process (assert_bucket.rs:75-77)
    cp = _build(tmp_path, Program(Num2Bits(16)), "num2bits16")
    r = ref_build.run_cli(cp, '{"in": "65536"}', tmp_path / "x.wtns")
    assert r.returncode != 0 and "Failed assert" in r.stdout


@pytest.mark.gpu
def test_gpu_wtns_equal_reference_runtime_wtns(tmp_path):
    from circom_amd import runtime as rt
    cp = compile_program(Program(Poseidon(2)), str(tmp_path), "poseidon2", sym=False)
    cli, loop = ref_build.binaries("bn128", "poseidon2")
    if not loop.exists():
        pytest.skip("oracle/_ref/bn128/poseidon2_loop not prebuilt")
    q = cp.flat.fp.q
    n = 512
    ins = _rand_inputs(q, n, 2, 21)
    raw = b"".join(v.to_bytes(32, "little") for row in ins for v in row)
    pre = str(tmp_path / "ref_")
    ref_build.run_loop(cp, raw, n, 1, wtns_prefix=pre, stride=8)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    b = c.batch(n)
    b.set_inputs(ins)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i in range(0, n, 8):
        g = tmp_path / ("gpu_%d.wtns" % i)
        b.write_wtns(i, g)
        assert g.read_bytes() == open(pre + "%d.wtns" % i, "rb").read(), i
    b.close(); c.close()


def test_json_value_forms_agree_with_the_reference_cli(tmp_path, ref_dir_bn128):
    """loadJson / json2FrElements (main.cpp:144-188,243-286): every way of writing a number — decimal, 0x, 0b, 0o
    strings, JSON integers, negative and fractional JSON numbers (which go through `double`), values above q —
    gives the same field element in the reference's CLI and in cw_set_inputs_json; what one rejects the other rejects."""
    from circom_amd import runtime as rt
    cp = _build(tmp_path, Program(Multiplier2()), "multiplier2")
    q = cp.flat.fp.q
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    forms = ['"12"', '"0x1F"', '"0xff"', '"0b101"', '"0o17"', '7', '-3', '0', '-0', '1e3', '1E2', '9007199254740993', '1.5', '2.5',
             '3.49', '-0.4', '-7.5', '1e30', '123456789012345678901234567890', '"%d"' % (q + 5), '"%d"' % (2 ** 300),
             '"%d"' % (q - 1), '"00012"', '"12a"', '"-1"', '""', 'true', 'null', '"0x"', '"1 2"', '[1]']
    b = c.batch(len(forms), device=-1)
    for i, form in enumerate(forms):
        text = '{"a": %s, "b": "1"}' % form
        out = tmp_path / ("j%d.wtns" % i)
        r = ref_build.run_cli(cp, text, out)
        try:
            b.set_inputs_json(i, text)
            mine = b.staged_input(i, 0)
        except rt.CwError:
            mine = None
        if r.returncode == 0:
            w = out.read_bytes()
            ref = int.from_bytes(w[76 + 64:76 + 96], "little")           # values start at byte 76; witness = [1, c, a, b]
            assert mine == ref, (form, mine, ref)
        else:
            assert mine is None, (form, mine, r.stdout[-200:], r.stderr[-200:])
    b.close(); c.close()
