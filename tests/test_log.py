"""log(...) statements (LogBucket: compiler/src/intermediate_representation/log_bucket.rs:105-162).

The emitted calculator prints every argument through printf (values with Fr_element2str, one blank between arguments, a
newline per statement).  Here the flat program carries one LOG row per argument, the lowering keeps every logged value in
hidden table slots behind the circuit's signals, and cw_get_log formats what the reference binary prints for ONE instance
of the batch.  tests/golden/reference_logs.json holds the reference CLI's stdout (tests/golden/make_golden.py logs)."""
import hashlib
import json
import os

import numpy as np
import pytest

from circom_amd import opcodes as O
from circom_amd.circuits.basic import LogDemo
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.frontend.flatten import flatten, log_program
from circom_amd.hip_elements.writers import wtns_bytes
from oracle.tape_eval import eval_flat, eval_tape

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_logs.json")))["cases"]["logdemo"]["vectors"]


def _oracle(fc, row):
    inp = {fc.main_input_start + k: int(row[n]) for k, n in enumerate(("a", "b"))}
    lines = []
    sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp, fc.functions, lines, fc.log_strings)
    return sig, failed, "".join(lines)


def test_flat_program_carries_one_row_per_argument():
    fc = flatten(Program(LogDemo()))
    op = fc.code["op"]
    assert (op == O.LOG).sum() == 17 and fc.n_log_values == 8
    stmts = log_program(fc)
    assert [len(items) for _, items in stmts] == [3, 4, 1, 0, 2, 3, 3]
    # the sub-component's statement sits where the component fires: after main stored its last input, before main goes on
    assert [fc.log_strings[i[1]] for _, items in stmts for i in items if i[0] == "s"] == \
        ["inputs:", "square of", "is", "constant", "out =", "(after the check)", "100%% of", "checks passed"]


def test_oracle_prints_what_the_reference_binary_prints():
    fc = flatten(Program(LogDemo()))
    for v in GOLD:
        sig, failed, text = _oracle(fc, v["inputs"])
        assert text == v["log"]
        assert (failed is None) == v["ok"]
        if v["ok"]:
            assert hashlib.sha256(wtns_bytes(fc.fp.q, sig)).hexdigest() == v["wtns_sha256"]


def test_lowered_schedule_keeps_logged_values_in_hidden_signals(tmp_path):
    from circom_amd import runtime as rt
    cp = compile_program(Program(LogDemo()), str(tmp_path), "logdemo", sym=False)
    fc = cp.flat
    assert cp.tape.n_signals == fc.n_signals + 8 and cp.tape.n_witness == fc.n_signals
    for v in GOLD[:3]:
        inp = {fc.main_input_start: int(v["inputs"]["a"]), fc.main_input_start + 1: int(v["inputs"]["b"])}
        sig, st = eval_tape(cp.tape, inp)
        assert st == 0
        want = [int(x) for line in v["log"].split("\n") for x in line.split() if x.isdigit()]
        assert sig[fc.n_signals:fc.n_signals + 8] == want
    # the C ABI: the hidden slots are no signals of the circuit, the witness list does not name them
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert c.n_signals == fc.n_signals and c.n_witness == fc.n_signals
    assert rt.lib().cw_n_log_statements(c.h) == 7
    c.close()
    # no bit-plane program for a circuit that logs (the bit table has no place for field-sized log values)
    from circom_amd.compiler import lower_bitplane
    assert lower_bitplane(fc, bits=True) is None


def test_cwf_carries_the_log_statements(tmp_path):
    from circom_amd import cwf
    fc = flatten(Program(LogDemo()))
    cwf.write_cwf(tmp_path / "l.cwf", fc)
    back = cwf.read_cwf(tmp_path / "l.cwf")
    assert back.log_strings == fc.log_strings and (back.code["op"] == fc.code["op"]).all()
    from circom_amd.hip_elements.lower import lower
    t1, t2 = lower(fc), lower(back)
    assert t1.log_prog == t2.log_prog and (t1.rows == t2.rows).all()


def test_reference_runtime_prints_the_same_text(tmp_path):
    from oracle import ref_build
    if not ref_build.REF_ROOT.exists():
        pytest.skip("no reference tree")
    cp = compile_program(Program(LogDemo()), str(tmp_path), "logdemo", sym=False, strands=(1,))
    ref_build.build_circuit(cp)
    fc = cp.flat
    for a, b in ((5, 6), (13, 2)):
        r = ref_build.run_cli(cp, json.dumps({"a": str(a), "b": str(b)}), tmp_path / "o.wtns")
        _, failed, text = _oracle(fc, {"a": a, "b": b})
        assert (r.returncode == 0) == (failed is None)
        assert r.stdout.startswith(text) and (failed is None) == (r.stdout == text)


@pytest.mark.gpu
def test_gpu_log_of_every_instance(tmp_path):
    from circom_amd import runtime as rt
    for mont in (False, True):
        cp = compile_program(Program(LogDemo()), str(tmp_path / ("m%d" % mont)), "logdemo", sym=False, mont=mont)
        c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
        assert bool(c.montgomery) == mont
        rows = [v for v in GOLD] * 30
        b = c.batch(len(rows))
        b.set_inputs([[int(v["inputs"]["a"]), int(v["inputs"]["b"])] for v in rows])
        b.run(); b.check_r1cs(); b.sync()
        st = b.status()
        for i, v in enumerate(rows):
            assert (st[i] == 0) == v["ok"], i
            assert b.log(i) == v["log"], i
            if v["ok"] and i < 10:
                b.write_wtns(i, tmp_path / "w.wtns")
                assert hashlib.sha256((tmp_path / "w.wtns").read_bytes()).hexdigest() == v["wtns_sha256"]
        b.close(); c.close()
