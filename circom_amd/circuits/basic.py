"""Small circuits from the reference documentation (used as plumbing cases and golden fixtures)."""
from ..frontend.dsl import template


@template
def Multiplier2(c):
    # mkdocs/docs/getting-started/writing-circuits.md:19-28
    a = c.input("a")
    b = c.input("b")
    out = c.output("c")
    c.set(out, a * b)


@template
def Internal(c):
    # mkdocs/docs/circom-language/formats/constraints-json.md:31-38
    inp = c.input("in", 2)
    out = c.output("out")
    c.set(out, inp[0] * inp[1])


@template
def BasicMain(c):
    # mkdocs/docs/circom-language/formats/constraints-json.md:40-47
    inp = c.input("in", 2)
    out = c.output("out")
    comp = c.component("c", Internal())
    c.set(comp["in"][0], inp[0])
    c.set(comp["in"][1], inp[1] + 2 * inp[0] + 1)
    c.set(out, comp["out"])


@template
def IsZero(c):
    # mkdocs/docs/circom-language/basic-operators.md:134-145 (circomlib comparators.circom IsZero)
    inp = c.input("in")
    out = c.output("out")
    inv = c.signal("inv")
    c.hint(inv, c.select(inp.neq(0), 1 / inp, 0))
    c.set(out, -inp * inv + 1)
    c.enforce(inp * out, 0)


@template
def Num2Bits(c, n):
    # mkdocs/docs/circom-language/basic-operators.md:147-169 (circomlib bitify.circom Num2Bits)
    inp = c.input("in")
    out = c.output("out", n)
    lc1 = c.const(0)
    e2 = c.const(1)
    for i in range(n):
        c.hint(out[i], (inp >> i) & 1)
        c.enforce(out[i] * (out[i] - 1), 0)
        lc1 = lc1 + out[i] * e2
        e2 = e2 + e2
    c.enforce(lc1, inp)
