"""Random circuits written TWICE from one random description - as circom source text and as eDSL template bodies - must give
identical flat circuits (signal numbering through nested component arrays, constraint forms, operation order, constants) and
identical witnesses.  The family: a chain of templates T0..Tk, each with an input / output / intermediate array, quadratic
`<==` rows over rotated indices, an optional array of sub-components of an earlier template wired through loops, a division
hint checked by `===`, and compile-time arithmetic on its parameter."""
import random

import pytest

from circom_amd.frontend.circom_exec import program_from_text
from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from oracle.field import PRIMES
from oracle.tape_eval import check_r1cs, eval_flat

Q = PRIMES["bn128"]


def _describe(rng, depth):
    """[{n_sub, sub (index of an earlier template or None), sub_n, rot, c1, c2, hint}] for T0..T(depth-1)"""
    out = []
    for k in range(depth):
        sub = rng.randrange(k) if k and rng.random() < 0.8 else None
        out.append(dict(sub=sub, n_sub=rng.randint(1, 3), sub_n=rng.randint(2, 4), rot=rng.randint(1, 3),
                        c1=rng.choice([1, 2, 7, Q - 1, 1 << 64]), c2=rng.choice([0, 1, 5, Q - 3]), hint=rng.random() < 0.6))
    return out


def _text(desc, main_n):
    parts = ["pragma circom 2.0.0;"]
    for k, d in enumerate(desc):
        body = ["template T%d(n) {" % k, "    signal input in[n];", "    signal output out[n];", "    signal mid[n];"]
        if d["hint"]:
            body += ["    signal aux;"]
        body += ["    var c1 = %d;" % d["c1"], "    var acc = n * 3 + %d;" % d["c2"],
                 "    for (var i = 0; i < n; i++) {", "        mid[i] <== in[i] * in[(i + %d) %% n] + c1 * in[i];" % d["rot"], "    }"]
        if d["sub"] is not None:
            body += ["    component sub[%d];" % d["n_sub"],
                     "    for (var s = 0; s < %d; s++) {" % d["n_sub"],
                     "        sub[s] = T%d(%d);" % (d["sub"], d["sub_n"]),
                     "        for (var i = 0; i < %d; i++) {" % d["sub_n"],
                     "            sub[s].in[i] <== mid[(i + s) % n] + acc;", "        }", "    }",
                     "    for (var i = 0; i < n; i++) {",
                     "        out[i] <== mid[i] * sub[i %% %d].out[i %% %d] + %d;" % (d["n_sub"], d["sub_n"], d["c2"]), "    }"]
        else:
            body += ["    for (var i = 0; i < n; i++) {", "        out[i] <== mid[i] * in[i] + %d;" % d["c2"], "    }"]
        if d["hint"]:
            body += ["    aux <-- in[0] / (in[1 % n] + 1);", "    aux * (in[1 % n] + 1) === in[0];"]
        parts.append("\n".join(body + ["}"]))
    parts.append("component main = T%d(%d);" % (len(desc) - 1, main_n))
    return "\n\n".join(parts)


def _edsl(desc):
    made = []
    for k, d in enumerate(desc):
        def body(c, n, d=d):
            inp = c.input("in", n)
            out = c.output("out", n)
            mid = c.signal("mid", n)
            aux = c.signal("aux") if d["hint"] else None
            c1 = d["c1"]
            acc = (n * 3 + d["c2"]) % Q
            for i in range(n):
                c.set(mid[i], inp[i] * inp[(i + d["rot"]) % n] + c1 * inp[i])
            if d["sub"] is not None:
                subs = []
                for s in range(d["n_sub"]):
                    sc = c.component("sub", made[d["sub"]](d["sub_n"]), s)
                    subs.append(sc)
                    for i in range(d["sub_n"]):
                        c.set(sc["in"][i], mid[(i + s) % n] + acc)
                for i in range(n):
                    c.set(out[i], mid[i] * subs[i % d["n_sub"]]["out"][i % d["sub_n"]] + d["c2"])
            else:
                for i in range(n):
                    c.set(out[i], mid[i] * inp[i] + d["c2"])
            if d["hint"]:
                c.hint(aux, inp[0] / (inp[1 % n] + 1))
                c.enforce(aux * (inp[1 % n] + 1), inp[0])
        body.__name__ = "T%d" % k
        made.append(template(body))
    return made[-1]


@pytest.mark.parametrize("seed", range(12))
def test_random_component_trees_from_text_and_from_the_edsl_are_one_circuit(seed):
    rng = random.Random(1000 + seed)
    desc = _describe(rng, rng.randint(2, 5))
    main_n = rng.randint(2, 5)
    a = flatten(program_from_text(_text(desc, main_n)))
    b = flatten(Program(_edsl(desc)(main_n)))
    assert a.n_signals == b.n_signals and a.inputs == b.inputs and a.n_outputs == b.n_outputs
    assert a.constraints == b.constraints and a.constants == b.constants
    for k in a.code:
        assert len(a.code[k]) == len(b.code[k]) and (a.code[k] == b.code[k]).all(), k
    assert a.signal_names() == b.signal_names()
    vals = [rng.randrange(Q) for _ in range(main_n)]
    inp = {a.main_input_start + i: v for i, v in enumerate(vals)}
    sa, fa = eval_flat(Q, a.n_signals, a.n_temps, a.constants, a.code, inp)
    sb, fb = eval_flat(Q, b.n_signals, b.n_temps, b.constants, b.code, inp)
    assert fa is None and fb is None and sa == sb and check_r1cs(Q, a.constraints, sa) is None
