"""Two independent ways to run a circom FUNCTION must agree: called with literals it is evaluated by the compile-time
executor (circom_exec, Python integers); called with signals it is inlined into the component's rows or compiled to tier-2
bytecode (circom_rt) and run by the oracle's interpreters.  Random functions: scalar variables and a four-entry array,
assignments, `if` / `else` on run-time comparisons, bounded `while` loops whose trip count depends on an argument, array
accesses at known and at value-dependent indices, early returns."""
import random

import pytest

from circom_amd.frontend.circom_exec import program_from_text
from circom_amd.frontend.flatten import flatten
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat

Q = PRIMES["bn128"]
VARS = ["x", "y", "z"]


def _expr(rng, depth=2):
    if depth == 0 or rng.random() < 0.3:
        r = rng.random()
        if r < 0.35:
            return rng.choice(["a", "b"])
        if r < 0.7:
            return rng.choice(VARS)
        if r < 0.85:
            return "t[%d]" % rng.randrange(4)
        return str(rng.choice([0, 1, 2, 3, 5, 7, 11]))
    op = rng.choice(["+", "-", "*", "+", "*"])
    return "(%s %s %s)" % (_expr(rng, depth - 1), op, _expr(rng, depth - 1))


def _cond(rng):
    return "(%s %% 7) %s (%s %% 5)" % (_expr(rng, 1), rng.choice(["<", ">", "==", "!=", "<=", ">="]), _expr(rng, 1))


def _stmts(rng, depth, n, loop_id, inline_only=False):
    out = []
    for _ in range(n):
        r = rng.random()
        if inline_only:
            # only what a trace can if-convert: assignments and `if` / `else` on run-time conditions, known indices
            r = r * 0.5 if r < 0.68 else 0.7
            if 0.5 <= r < 0.68:
                r = 0.1
        if r < 0.35:
            out.append("%s = %s;" % (rng.choice(VARS), _expr(rng)))
        elif r < 0.5:
            out.append("t[%d] = %s;" % (rng.randrange(4), _expr(rng)))
        elif r < 0.6:
            out.append("t[(%s) %% 4] = %s;" % (_expr(rng, 1), _expr(rng, 1)))          # a value-dependent index
        elif r < 0.68:
            out.append("%s = t[(%s) %% 4] + 1;" % (rng.choice(VARS), _expr(rng, 1)))
        elif r < 0.85 and depth > 0:
            s = "if (%s) { %s }" % (_cond(rng), " ".join(_stmts(rng, depth - 1, rng.randint(1, 3), loop_id, inline_only)))
            if rng.random() < 0.6:
                s += " else { %s }" % " ".join(_stmts(rng, depth - 1, rng.randint(1, 2), loop_id, inline_only))
            out.append(s)
        elif r < 0.95 and depth > 0:
            loop_id[0] += 1
            c = "c%d" % loop_id[0]
            out.append("var %s = 0; while (%s < (%s) %% 4) { %s %s++; }" % (
                c, c, _expr(rng, 1), " ".join(_stmts(rng, depth - 1, rng.randint(1, 2), loop_id)), c))
        elif depth > 0:
            out.append("if (%s) { return %s; }" % (_cond(rng), _expr(rng, 1)))
        else:
            out.append("%s = %s;" % (rng.choice(VARS), _expr(rng, 1)))
    return out


def _function(rng, inline_only=False):
    body = _stmts(rng, 2, rng.randint(3, 6), [0], inline_only)
    return ("function f(a, b) {\n    var x = a; var y = b; var z = 1; var t[4] = [1, a, b, 2];\n    %s\n    return %s + t[0] + t[3];\n}\n"
            % ("\n    ".join(body), _expr(rng)))


@pytest.mark.parametrize("seed", range(150))
def test_compile_time_and_run_time_execution_agree(seed):
    rng = random.Random(500 + seed)
    fn = _function(rng)
    run_time = flatten(program_from_text(fn + "template T() { signal input a; signal input b; signal output o; o <-- f(a, b); }\n"
                                              "component main = T();"))
    for a, b in [(3, 7), (0, 0), (rng.randrange(50), rng.randrange(50)), (Q - 1, 2)]:
        known = flatten(program_from_text(fn + "template C() { signal input u; signal output o; o <== f(%d, %d) + 0 * u; }\n"
                                               "component main = C();" % (a, b)))
        want, failed = eval_flat(Q, known.n_signals, known.n_temps, known.constants, known.code, {known.main_input_start: 0})
        assert failed is None
        got, failed = eval_flat(Q, run_time.n_signals, run_time.n_temps, run_time.constants, run_time.code,
                                {run_time.main_input_start: a, run_time.main_input_start + 1: b}, functions=run_time.functions)
        assert failed is None and got[1] == want[1], (fn, a, b)
        if run_time.functions:
            # the bytecode alone, every register but the arguments poisoned: nothing is read before it is written
            from oracle.field import Field
            from oracle.tape_eval import run_function
            f0 = run_time.functions[0]
            regs = [None] * f0["n_regs"]
            regs[:2] = [a % Q, b % Q]
            assert f0["n_args"] == 2 and run_function(Field(Q), f0, regs, 0, run_time.constants)
            assert regs[f0["ret_base"]] == want[1]


@pytest.mark.parametrize("seed", range(80))
def test_if_converted_functions_agree_with_compile_time_execution(seed):
    """functions without loops, early returns or value-dependent indices are INLINED into the component: their run-time `if`s
    are if-converted (both arms traced, variables and array elements merged through selects)"""
    rng = random.Random(9000 + seed)
    fn = _function(rng, inline_only=True)
    run_time = flatten(program_from_text(fn + "template T() { signal input a; signal input b; signal output o; o <-- f(a, b); }\n"
                                              "component main = T();"))
    assert not run_time.functions
    for a, b in [(3, 7), (0, 0), (rng.randrange(50), rng.randrange(50)), (Q - 1, 2)]:
        known = flatten(program_from_text(fn + "template C() { signal input u; signal output o; o <== f(%d, %d) + 0 * u; }\n"
                                               "component main = C();" % (a, b)))
        want, failed = eval_flat(Q, known.n_signals, known.n_temps, known.constants, known.code, {known.main_input_start: 0})
        assert failed is None
        got, failed = eval_flat(Q, run_time.n_signals, run_time.n_temps, run_time.constants, run_time.code,
                                {run_time.main_input_start: a, run_time.main_input_start + 1: b})
        assert failed is None and got[1] == want[1], (fn, a, b)


@pytest.mark.parametrize("seed", [501, 507, 523, 540, 577, 611])
def test_reference_runtime_executes_the_fuzzed_bytecode(seed, tmp_path):
    """the REFERENCE runtime runs the function as C++ control flow over its own Fr_* calls (oracle/emit_ref_cpp.py prints the
    bytecode): its witness equals the oracle's"""
    import os
    from oracle import ref_build
    if not ref_build.REF_ROOT.exists():
        pytest.skip("the reference tree is absent")
    from circom_amd.compiler import compile_program
    from circom_amd.hip_elements.writers import wtns_bytes
    rng = random.Random(seed)
    fn = _function(rng)
    prog = program_from_text(fn + "template T() { signal input a; signal input b; signal output o; o <-- f(a, b); }\ncomponent main = T();")
    cp = compile_program(prog, str(tmp_path), "txt_fuzz_%d" % seed, sym=False, strands=(1,), fpjit=False)
    fc = cp.flat
    if not fc.functions:
        pytest.skip("this one was inlined")
    rows = [(3, 7), (0, 0), (rng.randrange(50), rng.randrange(50)), (Q - 1, 2), (rng.randrange(Q), rng.randrange(1000))]
    ref_build.build_circuit(cp)
    raw = b"".join(int(v).to_bytes(32, "little") for r in rows for v in r)
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=str(tmp_path / "w_"))
    for i, (a, b) in enumerate(rows):
        sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code,
                                {fc.main_input_start: a, fc.main_input_start + 1: b}, functions=fc.functions)
        assert failed is None
        assert (tmp_path / ("w_%d.wtns" % i)).read_bytes() == wtns_bytes(Q, sig)
