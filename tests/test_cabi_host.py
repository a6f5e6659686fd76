"""C-ABI library: loads, exports every symbol include/circom_amd.h declares, and the host-side logic
(circuit loading, JSON ingest semantics of main.cpp:144-286, input errors of calcwit.cpp:51-97) behaves
like the reference — no GPU needed (host-only batches)."""
import ctypes
import re
from pathlib import Path

import pytest

from circom_amd import runtime as rt
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program, template
from circom_amd.circuits.basic import Multiplier2, BasicMain
from circom_amd.circuits.poseidon import Poseidon
from oracle.field import PRIMES

ROOT = Path(__file__).resolve().parent.parent
Q = PRIMES["bn128"]


def test_library_exports_every_declared_symbol():
    rt.build_library()
    hdr = (ROOT / "include" / "circom_amd.h").read_text()
    names = set(re.findall(r"\b(cw_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    L = ctypes.CDLL(str(rt.LIB_PATH))
    for n in sorted(names):
        assert hasattr(L, n), "library does not export " + n
    assert set(rt._SIGS) == names, set(rt._SIGS) ^ names
    assert b"gfx950" in rt.lib().cw_version()


def test_load_reports_circuit_shape(tmp_path):
    cp = compile_program(Program(Poseidon(2)), str(tmp_path), "poseidon2")
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert (c.n_signals, c.n_witness, c.n_inputs, c.input_start, c.n_constraints) == (1108, 1108, 2, 2, 1105)
    assert c.q == Q and c.input_size("inputs") == (2, 2) and c.input_size("nope") == (None, None)
    st = cp.tape.stats
    # field multiplications per witness: 81 S-boxes x 3 products (2 Montgomery products each) + 65 x 9 MDS products
    # signals in Montgomery form (compiler.choose_mont): 243 products + 585 matrix terms, one Montgomery product each
    assert c.montgomery and c.n_mmul == st["mmul"] + st["madd"] + st["mulc"] + 2 * st["mul2"] + st["linsum_terms"] == 828
    c.close()
    with pytest.raises(rt.CwError):
        rt.Circuit(str(tmp_path / "missing.cwt"))
    # .dat optional: hash map rebuilt from the tape's own name table
    c = rt.Circuit(cp.tape_path)
    assert c.input_size("inputs") == (2, 2)
    c.close()


def test_json_ingest_grammar_and_errors(tmp_path):
    cp = compile_program(Program(BasicMain()), str(tmp_path), "basic")
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    b = c.batch(8, device=-1)
    # decimal / hex / binary / octal strings, JSON numbers, negative numbers (floor-mod), huge values
    b.set_inputs_json(0, '{"in": ["12", "0x1F"]}')
    assert (b.staged_input(0, 0), b.staged_input(0, 1)) == (12, 31)
    b.set_inputs_json(1, '{"in": ["0b101", "0o17"]}')
    assert (b.staged_input(1, 0), b.staged_input(1, 1)) == (5, 15)
    b.set_inputs_json(2, '{"in": [7, -3]}')
    assert (b.staged_input(2, 0), b.staged_input(2, 1)) == (7, Q - 3)
    big = Q + 5
    b.set_inputs_json(3, '{"in": ["%d", "%d"]}' % (big, 2 ** 300))
    assert (b.staged_input(3, 0), b.staged_input(3, 1)) == (5, 2 ** 300 % Q)
    # JSON numbers go through double (main.cpp:170-175): 2^53+1 is not representable
    b.set_inputs_json(4, '{"in": [9007199254740993, 1e3]}')
    assert (b.staged_input(4, 0), b.staged_input(4, 1)) == (9007199254740992, 1000)
    assert b.remaining_inputs(4) == 0 and b.remaining_inputs(5) == 2
    for text, msg in (('{"in": ["1"]}', "Not enough values"), ('{"in": ["1","2","3"]}', "Too many values"),
                      ('{"nope": ["1","2"]}', "Signal not found"), ('{"in": ["12a", "1"]}', "Invalid number"),
                      ('{"in": ["-1", "1"]}', "Invalid number"), ('{"in": [true, 1]}', "Invalid JSON type"),
                      ('{"in": ["1", "2"]', "JSON parse error")):
        with pytest.raises(rt.CwError) as e:
            b.set_inputs_json(5, text)
        assert msg in str(e.value), (text, str(e.value))
    b.set_input_signal(6, "in", 0, 1)
    with pytest.raises(rt.CwError) as e:
        b.set_input_signal(6, "in", 0, 1)
    assert "assigned twice" in str(e.value)
    with pytest.raises(rt.CwError) as e:
        b.set_input_signal(6, "in", 2, 1)
    assert "exceeds the size" in str(e.value)
    with pytest.raises(rt.CwError) as e:
        b.set_input_signal(6, "in", 1, Q)
    assert "not reduced" in str(e.value)
    b.set_input_signal(6, "in", 1, 2)
    with pytest.raises(rt.CwError) as e:
        b.set_input_signal(6, "in", 1, 2)
    assert "No more signals" in str(e.value)
    # no GPU here: computing must fail loudly, never fall back
    with pytest.raises(rt.CwError) as e:
        b.run()
    assert "no CPU fallback" in str(e.value)
    b.close(); c.close()


@template
def BusLike(c):
    # inputs named like the flattened form of nested JSON objects (qualify_input, main.cpp:221-241)
    a = c.input("p.x")
    b_ = c.input("p.y", 2)
    d = c.input("pts[0].v")
    e = c.input("pts[1].v")
    out = c.output("out")
    c.set(out, a + b_[0] + b_[1] + d + e)


def test_json_nested_objects_are_qualified(tmp_path):
    cp = compile_program(Program(BusLike()), str(tmp_path), "bus")
    c = rt.Circuit(cp.tape_path, cp.dat_path, None)
    b = c.batch(1, device=-1)
    b.set_inputs_json(0, '{"p": {"x": 1, "y": [2, 3]}, "pts": [{"v": 4}, {"v": 5}]}')
    assert [b.staged_input(0, k) for k in range(5)] == [1, 2, 3, 4, 5]
    b.close(); c.close()


def test_cli_exists_and_fails_loudly_without_a_gpu(tmp_path):
    """The process-level drop-in (csrc/cw_cli.cpp) is a client of the C ABI: on a box without a GPU it must report
    the device error and exit non-zero — there is no CPU fallback to fall into."""
    import subprocess
    import torch
    from circom_amd.compiler import compile_program
    from circom_amd.frontend.dsl import Program
    from circom_amd.circuits.basic import Multiplier2
    assert rt.CLI_PATH.exists()
    r = subprocess.run([str(rt.CLI_PATH)], capture_output=True, text=True)
    assert r.returncode == 2 and "Usage" in r.stderr
    if torch.cuda.is_available():
        pytest.skip("GPU present: the computing path is covered by the -m gpu tests")
    compile_program(Program(Multiplier2()), str(tmp_path), "multiplier2")
    (tmp_path / "in.json").write_text('{"a": "3", "b": "11"}')
    r = subprocess.run([str(rt.CLI_PATH), str(tmp_path / "multiplier2"), str(tmp_path / "in.json"), str(tmp_path / "o.wtns")],
                       capture_output=True, text=True)
    assert r.returncode == 2 and r.stderr.strip() and not (tmp_path / "o.wtns").exists()


def test_public_signal_count_follows_the_main_declaration(tmp_path):
    """`component main {public [a]} = Multiplier2()`: outputs are always public, `a` becomes public input (r1cs header
    nPubOut/nPubIn, r1cs_writer.rs:245-269); cw_n_public is what a multi-GPU job gathers per instance."""
    cp = compile_program(Program(Multiplier2()), str(tmp_path), "m2")
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert c.n_public == 1
    c.close()
    cp = compile_program(Program(Multiplier2(), public=["a"]), str(tmp_path), "m2pub")
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    assert c.n_public == 2
    c.close()
