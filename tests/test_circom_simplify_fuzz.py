"""--O1 against --O0 on circuits nobody wrote: token-level mutations of valid sources that still compile are flattened, simplified,
and evaluated on random inputs - wherever a witness exists, it satisfies the full system exactly when its reduction satisfies the
simplified one (mutated circuits often do NOT satisfy their constraints: the two systems must agree on that too)."""
import os
import random

from circom_amd.frontend.circom_exec import program_from_text
from circom_amd.frontend.circom_lang import CircomSyntaxError, Source, tokenize
from circom_amd.frontend.circom_simplify import simplify_o1
from circom_amd.frontend.dsl import CircuitError
from circom_amd.frontend.flatten import flatten
from oracle.tape_eval import check_r1cs, eval_flat
from tests.test_circom_frontend import DOCS_BASIC, LIB


def test_o1_and_o0_agree_on_mutated_circuits():
    read = lambda name, main: open(os.path.join(LIB, name)).read() + "\ncomponent main = %s;" % main
    # (no seed with value-dependent loops: a mutated bound makes the oracle walk its 2^24-step limit)
    seeds = [DOCS_BASIC, read("bitify.circom", "Num2Bits(8)"), read("gates.circom", "MultiAND(5)"),
             read("multiplexer.circom", "Multiplexer(2, 3)"), read("switcher.circom", "Switcher()"),
             read("comparators.circom", "LessThan(6)")]
    rng = random.Random(11)
    compiled = satisfied = violated = 0
    for _ in range(1200):
        src = rng.choice(seeds)
        toks = tokenize(Source(0, "s", src))[:-1]
        spans = [t[2] for t in toks] + [len(src)]
        k = rng.randrange(len(toks))
        a, b = spans[k], spans[k + 1]
        op = rng.randrange(3)
        if op == 0:
            mut = src[:a] + src[b:]
        elif op == 1:
            j = rng.randrange(len(toks))
            mut = src[:a] + src[spans[j]:spans[j + 1]] + " " + src[b:]
        else:
            mut = src[:a] + rng.choice(["0", "1", "2", "in", "-", "+", "*", "x"]) + " " + src[b:]
        try:
            fc = flatten(program_from_text(mut, inspect=True))
        except (CircomSyntaxError, CircuitError):
            continue
        compiled += 1
        sm = simplify_o1(fc)
        assert set(range(1 + fc.n_outputs)) <= set(sm.witness2signal)
        inp = {fc.main_input_start + i: rng.randrange(fc.fp.q) if rng.random() < 0.5 else rng.randrange(4) for i in range(fc.n_main_inputs)}
        sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp, functions=fc.functions)
        if failed is not None:
            continue
        full_ok = check_r1cs(fc.fp.q, fc.constraints, sig) is None
        reduced_ok = check_r1cs(fc.fp.q, sm.constraints, [sig[s] for s in sm.witness2signal]) is None
        assert full_ok == reduced_ok, mut
        satisfied += full_ok
        violated += not full_ok
    assert compiled > 40 and satisfied > 15 and violated >= 0
