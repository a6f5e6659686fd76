"""`var` arrays of different lengths: variables are not strict (program_structure/src/utils/memory_slice.rs:129-160): the
overlapping positions are assigned, the rest keeps its values, and the compiler warns (execute.rs:3949-3965).  Signals stay
strict.  circom-ecdsa leans on this (its functions return `var out[100]` into whatever the caller declared)."""
import pytest

from circom_amd.frontend.circom_exec import program_from_text
from circom_amd.frontend.dsl import CircuitError
from circom_amd.frontend.flatten import flatten
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat

Q = PRIMES["bn128"]


def test_smaller_and_greater_arrays_into_variables():
    src = """
    function three() { var r[3] = [7, 8, 9]; return r; }
    function five() { var r[5] = [1, 2, 3, 4, 5]; return r; }
    template T() { signal input x; signal output o[6];
        var a[5] = [10, 20, 30, 40, 50];
        a = three();                       // smaller: a = [7, 8, 9, 40, 50]
        var b[2] = five();                 // greater: b = [1, 2]
        var m[2][3];
        m[1] = [x, x + 1];                 // one row, two of its three positions
        o[0] <== a[0] + a[3]; o[1] <== a[2] + a[4]; o[2] <== b[0] + b[1];
        o[3] <== m[1][0]; o[4] <== m[1][1]; o[5] <== m[1][2] + m[0][0]; }
    component main = T();"""
    prog = program_from_text(src)
    fc = flatten(prog)
    sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start: 100})
    assert failed is None and sig[1:7] == [47, 59, 3, 100, 101, 0]
    w = prog.world.typing_warnings
    assert len(w) == 3 and "smaller length, the remaining positions are not modified" in w[0] and "Expected length: 5, given 3" in w[0]
    assert "greater length" in w[1] and "Expected length: 2, given 5" in w[1]


def test_signals_stay_strict():
    with pytest.raises(CircuitError, match="different sizes"):
        program_from_text("template T() { signal input a[2]; signal output o[3]; o <== a; } component main = T();")


def test_the_same_inside_a_function_compiled_to_bytecode():
    # circom-ecdsa's habit: functions work on over-allocated arrays and hand them to differently sized variables
    src = """
    function big(x) { var r[6]; for (var i = 0; i < 6; i++) { r[i] = x + i; } return r; }
    function f(x) {
        var a[4] = big(x);                 // greater: a = x .. x + 3
        var b[6] = [9, 9, 9, 9, 9, 9];
        if (x > 5) { b = a; }              // run-time branch, smaller into a pinned array: b = a[0..3] ++ [9, 9]
        var s = 0;
        for (var i = 0; i < 6; i++) { s += b[i] * (i + 1); }
        var t = 0;
        while (t * t < x) { t++; }         // a trip count that depends on the value: this function becomes bytecode
        return s + t;
    }
    template T() { signal input x; signal output o; o <-- f(x); }
    component main = T();"""
    fc = flatten(program_from_text(src))
    assert len(fc.functions) == 1
    for x, want in ((3, 9 * 21 + 2), (10, 10 * 1 + 11 * 2 + 12 * 3 + 13 * 4 + 9 * 5 + 9 * 6 + 4)):
        sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start: x}, functions=fc.functions)
        assert failed is None and sig[1] == want
