"""Bus-typed signals in the circom-text front-end: every signal field is a signal of its own under its qualified name - the
name the `.sym` file, the input list of the `.dat` and `--inspect` use (`main.p[1].x[0]`) - and a bus named in main's public
list stands for all its fields (compiler/src/circuit_design/build.rs:300-425, constraint_writers sym_writer.rs)."""
from circom_amd.frontend.circom_exec import program_from_text
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements import writers

SRC = """pragma circom 2.2.0;
bus Point(n) { signal x[n]; signal y; }
template Main() { signal input k; input Point(2) p[2]; signal input z; signal output o; Point(1) mid;
  mid.x[0] <== p[1].x[1] * k; mid.y <== z;
  o <== mid.x[0] + mid.y + p[0].y; }
component main {public [p, z]} = Main();
"""


def test_sym_names_public_buses_and_input_list(tmp_path):
    prog = program_from_text(SRC, inspect=True)
    fc = flatten(prog)
    writers.write_sym(tmp_path / "b.sym", fc)
    assert [l.split(",")[3] for l in (tmp_path / "b.sym").read_text().split()] == [
        "main.o", "main.p[0].x[0]", "main.p[0].x[1]", "main.p[0].y", "main.p[1].x[0]", "main.p[1].x[1]", "main.p[1].y", "main.z",
        "main.k", "main.mid.x[0]", "main.mid.y"]
    # public inputs first (the bus p = six signals, then z), the private input k behind them; the whole bus once more at the end
    assert (fc.n_pub_in, fc.n_prv_in) == (7, 1)
    assert fc.inputs == [("p[0].x", 2, 2), ("p[0].y", 4, 1), ("p[1].x", 5, 2), ("p[1].y", 7, 1), ("z", 8, 1), ("k", 9, 1), ("p", 2, 6)]
    # --inspect names the fields that no constraint mentions
    assert prog.world.warnings == [
        'In template "Main()": Array of local signals p[0].x contains a total of 2 signals that do not appear in any constraint'
        ' = For example: p[0].x[0], p[0].x[1].',
        'In template "Main()": Local signal p[1].x[0] does not appear in any constraint',
        'In template "Main()": Local signal p[1].y does not appear in any constraint']
