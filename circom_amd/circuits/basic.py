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


@template
def PowerSums(c, n, m):
    """out[j][k] = in[j]^(k+1) for j < n, k < m: a template with a 2-dimensional output (its size depends on BOTH
    parameters), used as the element of a Mixed component array below"""
    x = c.input("in", n)
    out = c.output("out", n, m)
    for j in range(n):
        c.set(out[j][0], x[j] + 0)
        for k in range(1, m):
            c.set(out[j][k], out[j][k - 1] * x[j])


@template
def MixedArray(c, widths):
    """`component ps[len(widths)]; ps[i] = PowerSums(widths[i][0], widths[i][1]);` - one template NAME, different
    parameters per element: a `Mixed` cluster (compiler/src/intermediate_representation/translate.rs:1017-1045).  The
    reference addresses the signals of such components through the io-map it reads from the `.dat`
    (store_bucket.rs:498-566, load_bucket.rs:264-330) and runs them through `_functionTable`
    (store_bucket.rs:706-710): the parity case of SURVEY row a13's Mapped locations."""
    n_in = sum(w[0] for w in widths)
    x = c.input("x", n_in)
    s = c.output("s")
    ps = [c.component("ps", PowerSums(w[0], w[1]), index=i) for i, w in enumerate(widths)]
    k0 = 0
    acc = c.const(0)
    for i, (n, m) in enumerate(widths):
        for j in range(n):
            c.set(ps[i]["in"][j], x[k0 + j])
        k0 += n
    for i, (n, m) in enumerate(widths):
        for j in range(n):
            acc = acc + ps[i]["out"][j][m - 1] * (i + 2)
    c.set(s, acc)


@template
def LogSquare(c):
    """a sub-component that logs while it runs: its line appears where the component FIRES (after its last input)"""
    x = c.input("in")
    y = c.output("out")
    c.set(y, x * x)
    c.log("square of", x, "is", y)


@template
def LogDemo(c):
    """`log(...)` in the shapes LogBucket knows (log_bucket.rs:105-162): strings, signals, expressions, several arguments,
    no argument, inside a sub-component, before and after a run-time check that an instance may fail"""
    a = c.input("a")
    b = c.input("b")
    out = c.output("out")
    c.log("inputs:", a, b)
    sq = c.component("sq", LogSquare())
    c.set(sq["in"], a + b)
    c.log(a * b + 7)
    c.log()
    c.log("constant", 42)
    c.assert_(a.neq(13))
    c.set(out, sq["out"] + a)
    c.log("out =", out, "(after the check)")
    c.log("100%% of", 2, "checks passed")                       # the string is a printf format in the reference: prints "100% of"

