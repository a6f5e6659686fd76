"""integration/ holds the Rust side of the `--hip` target as SOURCE (no cargo in this environment: SURVEY f1).  What can be
checked without a compiler: the files are lexically well-formed, and the tables they share with the Python side - the `.cwf`
header fields, the code columns, the opcode numbering, the operand kinds - name the same things in the same order."""
import re
from pathlib import Path

from circom_amd import opcodes as O
from circom_amd import cwf

ROOT = Path(__file__).resolve().parent.parent
MOD = (ROOT / "integration" / "code_producers" / "src" / "hip_elements" / "mod.rs").read_text()
BACK = (ROOT / "integration" / "compiler" / "src" / "hip_backend.rs").read_text()


def _strip(src):
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r'b?"(?:\\.|[^"\\])*"', '""', src)
    return re.sub(r"'(?:\\.|[^'\\])'", "' '", src)


def test_rust_sources_are_lexically_balanced():
    for name, src in (("mod.rs", MOD), ("hip_backend.rs", BACK)):
        s = _strip(src)
        stack = []
        pairs = {")": "(", "]": "[", "}": "{"}
        for ch in s:
            if ch in "([{":
                stack.append(ch)
            elif ch in ")]}":
                assert stack and stack.pop() == pairs[ch], name
        assert not stack, name


def test_cwf_tables_agree_with_the_python_side():
    hdr = re.search(r"CWF_HEADER_FIELDS: \[&str; 10\] = \[(.*?)\];", MOD, re.S).group(1)
    fields = re.findall(r'"(\w+)"', hdr)
    doc = cwf.__doc__
    want = re.search(r"10 x u32: (.*?)\n\s+constants", doc, re.S).group(1)
    assert fields == [w.strip() for w in want.replace("\n", " ").split(",")]
    cols = re.findall(r'"(\w+)"', re.search(r"CWF_CODE_COLUMNS: \[&str; 9\] = \[(.*?)\];", MOD).group(1))
    assert tuple(cols) == cwf.COLS
    assert 'b"CWFL"' in MOD and cwf.MAGIC == b"CWFL"
    assert int(re.search(r"CWF_VERSION: u32 = (\d+)", MOD).group(1)) == cwf.VERSION
    # the header the Rust writer emits, in order
    body = re.search(r"let header: \[u32; 10\] = \[(.*?)\];", MOD, re.S).group(1)
    emitted = [re.sub(r"self\.|\.len\(\)| as u32|\s", "", x) for x in body.split(",") if x.strip()]
    assert emitted == ["total_number_of_signals", "n_temps", "field_tracking", "main_signal_offset", "number_of_main_inputs",
                       "number_of_public_inputs", "number_of_main_outputs", "main_input_list", "ops", "constraints"]


def test_opcode_numbering_agrees():
    enum = re.search(r"pub enum FlatOpcode \{(.*?)\n\}", MOD, re.S).group(1)
    nums = {k: int(v) for k, v in re.findall(r"(\w+) = (\d+),", enum)}
    want = {"Copy": O.COPY, "Add": O.ADD, "Sub": O.SUB, "Mul": O.MUL, "Div": O.DIV, "IntDiv": O.IDIV, "Mod": O.MOD, "Pow": O.POW,
            "Neg": O.NEG, "ShiftL": O.SHL, "ShiftR": O.SHR, "BitAnd": O.BAND, "BitOr": O.BOR, "BitXor": O.BXOR, "Complement": O.BNOT,
            "Lesser": O.LT, "Greater": O.GT, "LesserEq": O.LEQ, "GreaterEq": O.GEQ, "Eq": O.EQ, "NotEq": O.NEQ, "BoolAnd": O.LAND,
            "BoolOr": O.LOR, "BoolNot": O.LNOT, "Select": O.SELECT, "AssertEq": O.ASSERT_EQ, "AssertNz": O.ASSERT_NZ, "Run": O.RUN,
            "Call": O.CALL, "Log": O.LOG}
    assert nums == want
    kinds = {k: int(v) for k, v in re.findall(r"pub const (K_\w+): i64 = (\d+);", MOD)}
    assert kinds == {"K_SIG": O.K_SIG, "K_TMP": O.K_TMP, "K_CONST": O.K_CONST, "K_NONE": O.K_NONE}
    # every OperatorType of the reference's ComputeBucket has an arm in both tables of the trace
    ops = ["Mul", "Div", "Add", "Sub", "Pow", "IntDiv", "Mod", "ShiftL", "ShiftR", "LesserEq", "GreaterEq", "Lesser", "Greater",
           "Eq(_)", "NotEq", "BoolOr", "BoolAnd", "BitOr", "BitAnd", "BitXor", "PrefixSub", "BoolNot", "Complement"]
    for fn in ("fn flat_opcode", "fn fold"):
        body = BACK[BACK.index(fn):]
        body = body[:body.index("\n}\n")]
        for o in ops:
            assert re.search(r"\b%s =>" % re.escape(o), body), (fn, o)


def test_every_reference_accessor_the_rust_side_uses_exists():
    """the sketch cannot be compiled here, but what it reads from the reference's own types can be checked against the
    reference tree where that is present (the driver's container): fields / methods of `CProducer`, `Circuit`, `TemplateCode`"""
    import pytest
    ref = Path("/root/reference")
    if not ref.exists():
        pytest.skip("reference tree not present")
    cprod = (ref / "code_producers" / "src" / "c_elements" / "mod.rs").read_text()
    circuit = (ref / "compiler" / "src" / "circuit_design" / "circuit.rs").read_text()
    template = (ref / "compiler" / "src" / "circuit_design" / "template.rs").read_text()
    src = _strip(BACK) + _strip((ROOT / "integration" / "edits.md").read_text())
    for name in set(re.findall(r"\bc_producer\.(\w+)", src)):
        assert re.search(r"pub (fn )?%s\b" % name, cprod), "CProducer has no `%s`" % name
    for name in set(re.findall(r"\b(?:self|circuit)\.circuit\.(\w+)|\bcircuit\.(\w+)\(", src)):
        name = name[0] or name[1]
        if name in ("produce_hip",):                      # the trait this repository adds
            continue
        assert re.search(r"pub (fn )?%s\b" % name, circuit), "compiler::Circuit has no `%s`" % name
    for name in ("header", "number_of_inputs", "number_of_components", "body"):
        assert re.search(r"pub %s\b" % name, template), name
    assert "main_header_id" not in BACK and "constraints_for_hip" not in src
