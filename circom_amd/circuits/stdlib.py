"""Small building blocks in the shape of circomlib's comparators.circom, gates.circom, bitify.circom, binsum.circom,
switcher.circom and mux1.circom (circomlib itself is absent from the reference tree; re-authored from the
definitions).  Used by the parity tests as further witness-code shapes: comparisons through Num2Bits of a shifted
difference, boolean gates as degree-2 polynomials, bit recomposition, multi-operand binary sums."""
from ..frontend.dsl import template
from .basic import IsZero, Num2Bits


@template
def IsEqual(c):
    inp = c.input("in", 2)
    out = c.output("out")
    isz = c.component("isz", IsZero())
    c.set(isz["in"], inp[1] - inp[0])
    c.set(out, isz["out"])


@template
def LessThan(c, n):
    assert n <= 252
    inp = c.input("in", 2)
    out = c.output("out")
    n2b = c.component("n2b", Num2Bits(n + 1))
    c.set(n2b["in"], inp[0] + (1 << n) - inp[1])
    c.set(out, 1 - n2b["out"][n])


@template
def LessEqThan(c, n):
    inp = c.input("in", 2)
    out = c.output("out")
    lt = c.component("lt", LessThan(n))
    c.set(lt["in"][0], inp[0])
    c.set(lt["in"][1], inp[1] + 1)
    c.set(out, lt["out"])


@template
def GreaterThan(c, n):
    inp = c.input("in", 2)
    out = c.output("out")
    lt = c.component("lt", LessThan(n))
    c.set(lt["in"][0], inp[1])
    c.set(lt["in"][1], inp[0])
    c.set(out, lt["out"])


@template
def GreaterEqThan(c, n):
    inp = c.input("in", 2)
    out = c.output("out")
    lt = c.component("lt", LessThan(n))
    c.set(lt["in"][0], inp[1])
    c.set(lt["in"][1], inp[0] + 1)
    c.set(out, lt["out"])


@template
def XOR(c):
    a = c.input("a"); b = c.input("b"); out = c.output("out")
    c.set(out, a + b - 2 * a * b)


@template
def AND(c):
    a = c.input("a"); b = c.input("b"); out = c.output("out")
    c.set(out, a * b)


@template
def OR(c):
    a = c.input("a"); b = c.input("b"); out = c.output("out")
    c.set(out, a + b - a * b)


@template
def NOT(c):
    inp = c.input("in"); out = c.output("out")
    c.set(out, 1 + inp - 2 * inp)


@template
def Bits2Num(c, n):
    inp = c.input("in", n)
    out = c.output("out")
    lc = c.const(0)
    e2 = 1
    for i in range(n):
        lc = lc + inp[i] * e2
        e2 = e2 + e2
    c.set(out, lc)


@template
def Switcher(c):
    sel = c.input("sel"); L = c.input("L"); R = c.input("R")
    outL = c.output("outL"); outR = c.output("outR")
    aux = c.signal("aux")
    c.set(aux, (R - L) * sel)
    c.set(outL, aux + L)
    c.set(outR, -aux + R)


@template
def Mux1(c):
    cc = c.input("c", 2); s = c.input("s"); out = c.output("out")
    c.set(out, (cc[1] - cc[0]) * s + cc[0])


def _nbits(a):
    n, r = 1, 0
    while n - 1 < a:
        r += 1
        n *= 2
    return r


@template
def BinSum(c, n, ops):
    nout = _nbits(((1 << n) - 1) * ops)
    inp = c.input("in", ops, n)
    out = c.output("out", nout)
    lin = c.const(0)
    e2 = 1
    for k in range(n):
        for j in range(ops):
            lin = lin + inp[j][k] * e2
        e2 = e2 + e2
    lout = c.const(0)
    e2 = 1
    for k in range(nout):
        c.hint(out[k], (lin >> k) & 1)
        c.enforce(out[k] * (out[k] - 1), 0)
        lout = lout + out[k] * e2
        e2 = e2 + e2
    c.enforce(lin, lout)


@template
def SortPair(c, n):
    """Worked composition: order two n-bit numbers (comparator + switcher), report equality and the binary sum of
    their bit decompositions."""
    inp = c.input("in", 2)
    lo = c.output("lo"); hi = c.output("hi"); eq = c.output("eq"); total = c.output("sum")
    gt = c.component("gt", GreaterThan(n))
    c.set(gt["in"][0], inp[0]); c.set(gt["in"][1], inp[1])
    sw = c.component("sw", Switcher())
    c.set(sw["sel"], gt["out"]); c.set(sw["L"], inp[0]); c.set(sw["R"], inp[1])
    c.set(lo, sw["outL"]); c.set(hi, sw["outR"])
    ie = c.component("ie", IsEqual())
    c.set(ie["in"][0], inp[0]); c.set(ie["in"][1], inp[1])
    c.set(eq, ie["out"])
    ba = c.component("ba", Num2Bits(n)); c.set(ba["in"], inp[0])
    bb = c.component("bb", Num2Bits(n)); c.set(bb["in"], inp[1])
    bs = c.component("bs", BinSum(n, 2))
    for k in range(n):
        c.set(bs["in"][0][k], ba["out"][k])
        c.set(bs["in"][1][k], bb["out"][k])
    b2n = c.component("b2n", Bits2Num(n + 1))
    for k in range(n + 1):
        c.set(b2n["in"][k], bs["out"][k])
    c.set(total, b2n["out"])
