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


def _proj_dbl(X, Y, Z):
    """dbl-2008-bbjlp on (X : Y : Z): 3M + 4S, two multiplication levels"""
    B = (X + Y) * (X + Y)
    C = X * X
    Dd = Y * Y
    E = C * A
    F = E + Dd
    H = Z * Z
    J = F - 2 * H
    return (B - C - Dd) * J, F * (E - Dd), F * J


def _proj_add_affine(X1, Y1, Z1, x2, y2):
    """add-2008-bbjlp with Z2 = 1"""
    B = Z1 * Z1
    C = X1 * x2
    Dd = Y1 * y2
    E = C * Dd * D
    F = B - E
    G = B + E
    T = (X1 + Y1) * (x2 + y2) - C - Dd
    return Z1 * F * T, Z1 * G * (Dd - C * A), F * G


@template
def ScalarMulBitsProj(c, n):
    """Same relation as ScalarMulBits — out = (sum e[i] 2^i) * P, every intermediate point an (affine) signal tied to its
    predecessors by the BabyAdd constraints — but the WITNESS code walks the ladder in projective coordinates and obtains
    the affine signals from inversions of the Z's, none of which feeds the ladder: the dependent chain of the witness
    computation is multiplications only, the inversions are mutually independent (the lowering batches them, Montgomery's
    trick).  The bit-serial ScalarMulBits needs one inversion per addition ON the chain (~1000 in a row for EdDSA)."""
    e = c.input("e", n)
    px = c.input("px"); py = c.input("py")
    ox = c.output("outx"); oy = c.output("outy")
    chk = c.component("check", BabyCheck())
    c.set(chk["x"], px); c.set(chk["y"], py)
    selx = c.signal("selx", n); sely = c.signal("sely", n)
    dblx = c.signal("dblx", n); dbly = c.signal("dbly", n)
    addx = c.signal("addx", n); addy = c.signal("addy", n)
    aux = c.signal("aux", n, 2, 4)            # beta, gamma, delta, tau of the doubling / the addition of step i

    def constrain(i, which, x1, y1, x2, y2, xo, yo):
        beta, gamma, delta, tau = (aux[i][which][j] for j in range(4))
        c.set(beta, x1 * y2)
        c.set(gamma, y1 * x2)
        c.set(delta, (-A * x1 + y1) * (x2 + y2))
        c.set(tau, beta * gamma)
        c.enforce((1 + D * tau) * xo, beta + gamma)
        c.enforce((1 - D * tau) * yo, delta + A * beta - gamma)

    X, Y, Z = c.const(0), c.const(1), c.const(1)
    accx, accy = c.const(0), c.const(1)
    for i in range(n - 1, -1, -1):
        c.enforce(e[i] * (e[i] - 1), 0)
        c.set(selx[i], e[i] * px)                    # e ? P : (0, 1)
        c.set(sely[i], e[i] * (py - 1) + 1)
        X, Y, Z = _proj_dbl(X, Y, Z)
        zi = 1 / Z
        c.hint(dblx[i], X * zi)
        c.hint(dbly[i], Y * zi)
        constrain(i, 0, accx, accy, accx, accy, dblx[i], dbly[i])
        X, Y, Z = _proj_add_affine(X, Y, Z, selx[i], sely[i])
        zi = 1 / Z
        c.hint(addx[i], X * zi)
        c.hint(addy[i], Y * zi)
        constrain(i, 1, dblx[i], dbly[i], selx[i], sely[i], addx[i], addy[i])
        accx, accy = addx[i], addy[i]
    c.set(ox, accx); c.set(oy, accy)
