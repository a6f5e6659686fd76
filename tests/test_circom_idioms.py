"""Idioms of real circom libraries that no eDSL circuit of this tree mirrors: a template that instantiates itself (MultiAND), array
signals initialised in their declaration, two-dimensional variables filled by compile-time functions, `-->` / `==>`, components
declared first and instantiated inside known branches."""
import os
import random

import pytest

from circom_amd.frontend.circom_exec import build_program, program_from_text
from circom_amd.frontend.circom_lang import parse_program
from circom_amd.frontend.flatten import flatten
from oracle.field import PRIMES
from oracle.tape_eval import check_r1cs, eval_flat

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "circom_amd", "circuits", "circomlib")
Q = PRIMES["bn128"]


def _run(fc, vals):
    sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code,
                            {fc.main_input_start + k: v % Q for k, v in enumerate(vals)}, functions=fc.functions)
    return sig, failed


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 11])
def test_recursive_template(tmp_path, n):
    f = tmp_path / "a.circom"
    f.write_text('include "gates.circom";\ncomponent main = MultiAND(%d);\n' % n)
    prog = build_program(parse_program(str(f), [LIB]))
    fc = flatten(prog)
    # n - 1 AND gates whatever the split
    assert sum(1 for a, b, c in fc.constraints if a) == max(0, n - 1)
    rng = random.Random(n)
    for bits in ([1] * n, [1] * (n - 1) + [0], [rng.randrange(2) for _ in range(n)]):
        sig, failed = _run(fc, bits)
        assert failed is None and sig[1] == int(all(bits)) and check_r1cs(Q, fc.constraints, sig) is None
    if n == 5:
        # MultiAND(5) -> ands[0] = MultiAND(2), ands[1] = MultiAND(3): one array, two parameter sets
        names = {(i.name, i.params) for i in prog.inst_list}
        assert {("MultiAND", (2,)), ("MultiAND", (3,)), ("MultiAND", (5,))} <= names


def test_declaration_forms_and_compile_time_tables():
    src = """
    function pascal(n) {                       // a two-dimensional variable filled at compile time
        var t[6][6];
        for (var i = 0; i < n; i++) {
            t[i][0] = 1;
            for (var j = 1; j <= i; j++) { t[i][j] = t[i - 1][j - 1] + t[i - 1][j]; }
        }
        return t;
    }
    template Row(n, r) {
        signal input x;
        var t[6][6] = pascal(n);
        signal output out[n] <== [t[r][0] * x, t[r][1] * x, t[r][2] * x, t[r][3] * x];   // initialised in the declaration
        signal acc <== out[1] * out[2];
        signal output prod;
        acc ==> prod;
        signal output hinted;
        x * 2 --> hinted;
        hinted === x + x;
    }
    component main = Row(4, 3);"""
    fc = flatten(program_from_text(src))
    sig, failed = _run(fc, [5])
    # outputs in declaration order: out[4], prod, hinted
    assert failed is None and sig[1:7] == [5, 15, 15, 5, 225, 10] and check_r1cs(Q, fc.constraints, sig) is None


def test_output_of_a_component_without_all_its_inputs_is_refused():
    """execute.rs:3973: `o <== c.z; c.x <== a;` - the reference refuses to read c.z before c has every input; reading an
    INPUT of the component, or the output once the inputs are in, stays allowed"""
    from circom_amd.frontend.dsl import CircuitError as CircomError
    sq = "template Sq() { signal input x; signal output z; z <== x*x; }\n"
    bad = sq + "template M() { signal input a; signal output o; component c = Sq(); o <== c.z; c.x <== a; }\ncomponent main = M();\n"
    with pytest.raises(CircomError, match="not all its inputs initialized"):
        flatten(program_from_text(bad))
    good = sq + "template M() { signal input a; signal output o; component c = Sq(); c.x <== a; o <== c.z + c.x; }\ncomponent main = M();\n"
    fc = flatten(program_from_text(good))
    sig, failed = _run(fc, [5])
    assert failed is None and sig[1] == 30
