"""BabyJubjub (twisted Edwards a=168700, d=168696 over the bn128 scalar field) point arithmetic in the shape of
circomlib's babyjub.circom, plus a bit-serial scalar multiplication (the core of EdDSA verification, BASELINE
config 4).  The coordinate formulas use `<--` with field division followed by a `===` check, so these circuits
drive the slow-path operators (DIV -> INV) of the schedule at scale.  Re-authored (circomlib absent); outputs
are pinned against plain-integer Edwards arithmetic in tests."""
from ..frontend.dsl import template

A = 168700
D = 168696
BASE8 = (5299619240641551281634865583518297030282874472190772894086521144482721001553,
         16950150798460657717958625567821834550301663161624707787222815936182638968203)


@template
def BabyAdd(c):
    x1 = c.input("x1"); y1 = c.input("y1"); x2 = c.input("x2"); y2 = c.input("y2")
    xout = c.output("xout"); yout = c.output("yout")
    beta = c.signal("beta"); gamma = c.signal("gamma"); delta = c.signal("delta"); tau = c.signal("tau")
    c.set(beta, x1 * y2)
    c.set(gamma, y1 * x2)
    c.set(delta, (-A * x1 + y1) * (x2 + y2))
    c.set(tau, beta * gamma)
    c.hint(xout, (beta + gamma) / (1 + D * tau))
    c.enforce((1 + D * tau) * xout, beta + gamma)
    c.hint(yout, (delta + A * beta - gamma) / (1 - D * tau))
    c.enforce((1 - D * tau) * yout, delta + A * beta - gamma)


@template
def BabyDbl(c):
    x = c.input("x"); y = c.input("y")
    xout = c.output("xout"); yout = c.output("yout")
    adder = c.component("adder", BabyAdd())
    c.set(adder["x1"], x); c.set(adder["y1"], y); c.set(adder["x2"], x); c.set(adder["y2"], y)
    c.set(xout, adder["xout"]); c.set(yout, adder["yout"])


@template
def BabyCheck(c):
    x = c.input("x"); y = c.input("y")
    x2 = c.signal("x2"); y2 = c.signal("y2")
    c.set(x2, x * x)
    c.set(y2, y * y)
    c.enforce(A * x2 + y2, 1 + D * x2 * y2)


@template
def ScalarMulBits(c, n):
    """out = (sum e[i] 2^i) * P by double-and-add from the most significant bit (e[i] are checked to be bits)."""
    e = c.input("e", n)
    px = c.input("px"); py = c.input("py")
    ox = c.output("outx"); oy = c.output("outy")
    chk = c.component("check", BabyCheck())
    c.set(chk["x"], px); c.set(chk["y"], py)
    selx = c.signal("selx", n); sely = c.signal("sely", n)
    accx, accy = c.const(0), c.const(1)
    for i in range(n - 1, -1, -1):
        c.enforce(e[i] * (e[i] - 1), 0)
        dbl = c.component("dbl", BabyDbl(), i)
        c.set(dbl["x"], accx); c.set(dbl["y"], accy)
        c.set(selx[i], e[i] * px)                    # e ? P : (0, 1)
        c.set(sely[i], e[i] * (py - 1) + 1)
        add = c.component("add", BabyAdd(), i)
        c.set(add["x1"], dbl["xout"]); c.set(add["y1"], dbl["yout"])
        c.set(add["x2"], selx[i]); c.set(add["y2"], sely[i])
        accx, accy = add["xout"], add["yout"]
    c.set(ox, accx); c.set(oy, accy)
