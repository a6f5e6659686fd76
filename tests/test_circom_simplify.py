"""`--O1`, the reference's DEFAULT simplification (constant and renaming substitutions only), restated in
circom_amd/frontend/circom_simplify.py from constraint_list/src/constraint_simplification.rs:

  * the documentation's own listings for basic.circom at the default level: constraints JSON
    (mkdocs formats/constraints-json.md:52-63) and `.sym` with -1 for the eliminated signals (formats/sym.md:49-60);
  * on larger circuits: the reduced witness satisfies the simplified system, every removed signal is a renamed one (equal
    to its representative in every witness), a constant or unused, outputs and public inputs always stay;
  * the REFERENCE runtime, given a `.dat` with the simplified witness list, writes exactly the bytes `reduce_wtns` cuts out of
    the full `.wtns`."""
import json
import os
import random
import struct

import pytest

from circom_amd.frontend.circom_exec import program_from_file, program_from_text
from circom_amd.frontend.circom_simplify import constraints_json, reduce_wtns, simplify_o1, write_r1cs, write_sym
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements.writers import wtns_bytes
from oracle.tape_eval import check_r1cs
from tests.test_circom_frontend import DOCS_BASIC, LIB, SRC, Q, libs, run  # noqa: F401  (libs is a fixture)

QM1 = str(Q - 1)


def test_docs_basic_circom_at_the_default_level(tmp_path):
    sm = simplify_o1(flatten(program_from_text(DOCS_BASIC)))
    # constraints-json.md:56-63 ("only constant and renaming simplifications have been applied, since --O1 is the default")
    assert json.loads(constraints_json(sm.constraints))["constraints"] == [
        [{"2": QM1}, {"4": "1"}, {"1": QM1}],
        [{}, {}, {"0": "1", "2": "2", "3": "1", "4": QM1}]]
    # sym.md:51-58 ("two signals have been eliminated")
    write_sym(tmp_path / "b.sym", sm)
    assert (tmp_path / "b.sym").read_text() == \
        "1,1,1,main.out\n2,2,1,main.in[0]\n3,3,1,main.in[1]\n4,-1,0,main.c.out\n5,-1,0,main.c.in[0]\n6,4,0,main.c.in[1]\n"
    assert sm.witness2signal == [0, 1, 2, 3, 6] and sm.substituted == {5: 2, 4: 1}
    # .r1cs: 5 wires, 7 labels, wire2label = the kept signals
    write_r1cs(tmp_path / "b.r1cs", sm)
    raw = (tmp_path / "b.r1cs").read_bytes()
    assert raw[:4] == b"r1cs"
    off = 12
    secs = {}
    while off < len(raw):
        typ, ln = struct.unpack_from("<IQ", raw, off)
        secs[typ] = raw[off + 12:off + 12 + ln]
        off += 12 + ln
    n_wires, n_out, n_pub, n_prv, n_labels, n_cons = struct.unpack_from("<IIIIQI", secs[1], 4 + 32)
    assert (n_wires, n_out, n_pub, n_prv, n_labels, n_cons) == (5, 1, 0, 2, 7, 2)
    assert list(struct.unpack("<5Q", secs[3])) == [0, 1, 2, 3, 6]


def _invariants(fc, sm, rows):
    q = fc.fp.q
    forbidden = {0} | set(range(1, 1 + fc.n_outputs)) | set(range(fc.main_input_start, fc.main_input_start + fc.n_pub_in))
    kept = set(sm.witness2signal)
    assert forbidden <= kept and sm.witness2signal == sorted(kept)
    assert kept | set(sm.substituted) | set(sm.constants) | set(sm.unused) == set(range(fc.n_signals))
    mentioned = set()
    for con in fc.constraints:
        for part in con:
            mentioned.update(part)
    for r in rows:
        sig, failed = run(fc, r)
        assert failed is None and check_r1cs(q, fc.constraints, sig) is None
        red = [sig[s] for s in sm.witness2signal]
        assert check_r1cs(q, sm.constraints, red) is None
        assert all(sig[s] == sig[rep] for s, rep in sm.substituted.items())
        assert all(sig[s] == v for s, v in sm.constants.items())
    # a corrupted witness entry that the simplified system still mentions is still caught
    sig, _ = run(fc, rows[0])
    red = [sig[s] for s in sm.witness2signal]
    used = sorted({k for con in sm.constraints for part in con for k in part if k})
    red[used[len(used) // 2]] = (red[used[len(used) // 2]] + 1) % q
    assert check_r1cs(q, sm.constraints, red) is not None


def test_simplified_systems_of_larger_circuits(libs):
    rng = random.Random(8)
    fc = flatten(program_from_file(os.path.join(SRC, "sortpair.circom"), libs))
    sm = simplify_o1(fc)
    _invariants(fc, sm, [[rng.getrandbits(16), rng.getrandbits(16)] for _ in range(4)] + [[9, 9]])
    assert (fc.n_signals, len(fc.constraints)) == (144, 145) and (sm.n_wires, len(sm.constraints)) == (78, 79)
    fc = flatten(program_from_file(os.path.join(SRC, "poseidon2.circom"), libs))
    sm = simplify_o1(fc)
    _invariants(fc, sm, [[rng.randrange(Q), rng.randrange(Q)] for _ in range(2)] + [[0, 0]])
    # Poseidon(2): the wiring between Ark / Sigma / Mix components disappears
    assert fc.n_signals == 1108 and sm.n_wires < 700 and len(sm.constraints) < 700
    fc = flatten(program_from_file(os.path.join(SRC, "mixed_array.circom"), libs))
    _invariants(fc, simplify_o1(fc), [[rng.randrange(Q) for _ in range(8)]])
    # one SHA-256 block: 204 329 signals of wiring at --O0, 31 017 wires / 31 264 constraints once renamings and constants are
    # gone (169 936 renamed signals, 3 376 constants) - the size the circuit is usually quoted with (~30 K per block)
    fc = flatten(program_from_file(os.path.join(SRC, "sha256_64.circom"), libs))
    sm = simplify_o1(fc)
    assert (sm.n_wires, len(sm.constraints), len(sm.substituted), len(sm.constants), len(sm.unused)) == (31017, 31264, 169936, 3376, 0)
    _invariants(fc, sm, [[rng.randrange(2) for _ in range(64)]])


def test_public_inputs_are_kept_private_ones_may_go():
    src = """template T() { signal input a; signal input b; signal input unused; signal output o; signal output p;
        signal m; m <== a; o <== m * b; p <== 7; }
    component main {public [b]} = T();"""
    fc = flatten(program_from_text(src))
    sm = simplify_o1(fc)
    names = fc.signal_names()
    kept = [names[s] for s in sm.witness2signal[1:]]
    # p = 7 stays as a constraint (an output is forbidden), m is renamed to a, the unused private input leaves the witness
    assert kept == ["main.o", "main.p", "main.b", "main.a"]
    assert sm.n_prv_in == 1 and fc.n_prv_in == 2
    cons = json.loads(constraints_json(sm.constraints))["constraints"]
    assert [{}, {}, {"0": "7", "2": QM1}] in cons or [{}, {}, {"0": str(Q - 7), "2": "1"}] in cons


def test_reference_runtime_writes_the_reduced_witness(tmp_path, libs):
    from oracle import ref_build
    if not ref_build.REF_ROOT.exists():
        pytest.skip("the reference tree is absent")
    from circom_amd.compiler import compile_program
    prog = program_from_file(os.path.join(SRC, "sortpair.circom"), libs)
    cp = compile_program(prog, str(tmp_path), "sortpair_o1src", sym=False, strands=(1,), fpjit=False)
    fc = cp.flat
    sm = simplify_o1(fc)
    cli = ref_build.build_cli_with_witness_list(cp, sm.witness2signal, "sortpair_o1")
    import subprocess
    for a, b in ((40000, 123), (5, 5), (0, 65535)):
        (tmp_path / "in.json").write_text(json.dumps({"in": [str(a), str(b)]}))
        r = subprocess.run([str(cli), str(tmp_path / "in.json"), str(tmp_path / "o1.wtns")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        sig, failed = run(fc, [a, b])
        full = wtns_bytes(fc.fp.q, sig)
        got = (tmp_path / "o1.wtns").read_bytes()
        assert got == reduce_wtns(full, sm.witness2signal)
        assert len(got) == 76 + 32 * sm.n_wires and got != full
