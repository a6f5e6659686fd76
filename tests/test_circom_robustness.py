"""The text front-end never answers a broken program with a Python exception, a hang or a crash: token-level mutations of valid
sources end in CircomSyntaxError / CircuitError (or are still valid programs), and a template that instantiates itself without
reaching a base case ends as an error instead of overflowing the interpreter's stack."""
import os
import random

import pytest

from circom_amd.frontend.circom_exec import program_from_text
from circom_amd.frontend.circom_lang import CircomSyntaxError, Source, tokenize
from circom_amd.frontend.dsl import CircuitError
from tests.test_circom_frontend import DOCS_BASIC, LIB, RT_SRC


def _read(name, main):
    return open(os.path.join(LIB, name)).read() + "\ncomponent main = %s;" % main


def test_token_mutations_end_in_front_end_errors():
    seeds = [DOCS_BASIC, RT_SRC, _read("bitify.circom", "Num2Bits(8)"), _read("gates.circom", "MultiAND(5)"),
             _read("multiplexer.circom", "Multiplexer(2, 3)"), _read("switcher.circom", "Switcher()")]
    rng = random.Random(7)
    still_valid = rejected = 0
    for _ in range(700):
        src = rng.choice(seeds)
        toks = tokenize(Source(0, "s", src))[:-1]
        spans = [t[2] for t in toks] + [len(src)]
        k = rng.randrange(len(toks))
        a, b = spans[k], spans[k + 1]
        op = rng.randrange(4)
        if op == 0:
            mut = src[:a] + src[b:]
        elif op == 1:
            mut = src[:b] + " " + src[a:b] + src[b:]
        elif op == 2:
            j = rng.randrange(len(toks))
            mut = src[:a] + src[spans[j]:spans[j + 1]] + " " + src[b:]
        else:
            mut = src[:a] + rng.choice(["0", "(", ")", "[", "]", "{", "}", ";", "<==", "in", "1 <<", "-", "x"]) + " " + src[b:]
        try:
            program_from_text(mut)
            still_valid += 1
        except (CircomSyntaxError, CircuitError):
            rejected += 1
    assert rejected > 500 and still_valid > 5


def test_runaway_template_recursion_is_an_error():
    with pytest.raises(CircuitError, match="nested more than 150 deep"):
        program_from_text("template T(n) { signal input a; signal output o; component c = T(n + 1); c.a <== a; o <== c.o; } "
                          "component main = T(0);")
    with pytest.raises(CircuitError, match="nested too deeply"):
        program_from_text("function f(n) { return f(n + 1); } template T() { signal input a; signal output o; o <== a * f(0); } "
                          "component main = T();")


def test_pathological_nesting_is_a_front_end_error():
    deep = "(" * 3000 + "a" + ")" * 3000
    with pytest.raises((CircomSyntaxError, CircuitError), match="nested too deeply"):
        program_from_text("template T() { signal input a; signal output o; o <== %s; }\ncomponent main = T();" % deep)
    # a LONG expression is not a deep one: 12 000 terms
    from circom_amd.frontend.flatten import flatten
    src = "template T() { signal input a; signal output o; o <== %s; }\ncomponent main = T();" \
          % " + ".join("a * %d" % (i % 7 + 1) for i in range(12000))
    assert len(flatten(program_from_text(src)).constraints) == 1
