"""Foreign-field elliptic-curve arithmetic and ECDSA verification in the shape of 0xPARC circom-ecdsa (`bigint.circom`,
`secp256k1.circom`, `ecdsa.circom`): BASELINE.json config 5 — secp256k1 signature verification inside a circuit over the
BLS12-381 scalar field, numbers as k limbs of n bits (n = 64, k = 4).

circom-ecdsa is not part of the reference tree; the templates are re-authored from its published structure, with two
simplifications that keep every gadget generic in the foreign prime:
  * the chord / tangent slope `lambda` is a (range-checked) SIGNAL and each curve operation is three QUADRATIC relations
    modulo p — lambda (x2 - x1) = y2 - y1, lambda^2 = x1 + x2 + x3, lambda (x1 - x3) = y1 + y3 — where circom-ecdsa folds
    them into one cubic relation and reduces its ten registers with the special form of the secp256k1 prime;
  * "X = 0 (mod p)" is one gadget, `CheckZeroModP`: X arrives as overflowed registers (limb convolutions `BigMultNoCarry`,
    possibly negative), a constant that is 0 (mod p) makes every register non-negative, the quotient q = X / p is a hint from
    the run-time function `long_div` (bigint_func.py), its limbs are range-checked, and X - q p is proved to carry to zero
    (`CheckCarryToZero`: one linear constraint and one range-checked carry per register).
The `<--` side is circom-ecdsa's: `secp256k1_addunequal_func` / `secp256k1_double_func` / `mod_inv` as circom FUNCTIONS with
run-time control flow (bigint_func.py) — a modular inverse by Fermat costs ~10^6 interpreted operations, which is why those
three carry a native closed form for the oracle and the device.

Templates take the curve as a parameter tuple `cv = (p, order, Gx, Gy)` so that the tests can run them on a toy curve.
"""
from ..frontend.dsl import template
from .basic import Num2Bits, IsZero
from .stdlib import LessThan, IsEqual
from . import bigint_func as BF

SECP256K1 = (2 ** 256 - 2 ** 32 - 977,
             0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
             0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
             0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)


# ---- host-side curve arithmetic (table constants, input synthesis, tests) -------------------------------------------------
def ec_add(cv, P, Q):
    p = cv[0]
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0:
            return None
        lam = 3 * P[0] * P[0] * pow(2 * P[1], p - 2, p) % p
    else:
        lam = (Q[1] - P[1]) * pow((Q[0] - P[0]) % p, p - 2, p) % p
    x = (lam * lam - P[0] - Q[0]) % p
    return x, (lam * (P[0] - x) - P[1]) % p


def ec_mul(cv, s, P):
    R = None
    while s:
        if s & 1:
            R = ec_add(cv, R, P)
        P = ec_add(cv, P, P)
        s >>= 1
    return R


# ---- big-integer gadgets ------------------------------------------------------------------------------------------------------
@template
def BigMultNoCarry(c, n, ka, kb):
    """out[ka + kb - 1] = the limb convolution of a and b, no carries (bigint.circom BigMultNoCarry): the registers are hints,
    pinned by the polynomial identity a(x) b(x) = out(x) at ka + kb - 1 points (one constraint each)"""
    a = c.input("a", ka)
    b = c.input("b", kb)
    m = ka + kb - 1
    out = c.output("out", m)
    for i in range(m):
        acc = c.const(0)
        for j in range(max(0, i - kb + 1), min(ka, i + 1)):
            acc = acc + a[j] * b[i - j]
        c.hint(out[i], acc)
    for x in range(m):
        pa = c.const(0)
        for j in range(ka):
            pa = pa + a[j] * (x ** j)
        pb = c.const(0)
        for j in range(kb):
            pb = pb + b[j] * (x ** j)
        po = c.const(0)
        for j in range(m):
            po = po + out[j] * (x ** j)
        c.enforce(pa * pb, po)


def _zero_offset(n, m, B, p):
    """constants C[m] with C_i >= 2^B and sum C_i 2^(n i) = 0 (mod p): added to registers |X_i| < 2^B they make every
    register non-negative without changing the value modulo p"""
    C = [1 << B] * m
    v = sum(ci << (n * i) for i, ci in enumerate(C)) % p
    r = (-v) % p
    for i, limb in enumerate(BF.limbs_of(r, n, (p.bit_length() + n - 1) // n)):
        C[i] += limb
    assert sum(ci << (n * i) for i, ci in enumerate(C)) % p == 0
    return C


@template
def CheckZeroModP(c, n, k, m, B, p):
    """in[m]: registers of an integer X = sum in_i 2^(n i), |in_i| < 2^B (negative values as field negatives).
    Proves X = 0 (mod p)."""
    inp = c.input("in", m)
    C = _zero_offset(n, m, B, p)
    # proper (n-bit limb) representation of X + C, by carry propagation in `<--` code
    tb = B + 3                                                    # in_i + C_i + carry < 2^tb
    extra = max(1, -(-(tb - n) // n))
    M = m + extra
    while M < k + 1:
        M += 1
    proper = []
    carry = c.const(0)
    for i in range(M):
        t = (inp[i] + C[i] + carry) if i < m else carry
        proper.append(t % (1 << n))
        carry = t // (1 << n)
    # the divisor is the constant prime (top limb != 0) and the dividend's limbs are proper by construction: the quotient /
    # remainder pair is unique, so the function carries its closed form (bigint_func.native_eval "long_div")
    fn = c.function("long_div_%d_%d_%d" % (n, k, M - k), M + k, BF.build_long_div(n, k, M - k),
                    native=("long_div", n, k, M - k) if BF.limbs_of(p, n, k)[k - 1] != 0 else None)
    res = c.call(fn, proper + BF.limbs_of(p, n, k))
    Lq = M - k + 1
    q = c.signal("q", Lq)
    for j in range(Lq):
        c.hint(q[j], res[j])
        rc = c.component("q_range", Num2Bits(n), j)
        c.set(rc["in"], q[j])
    # Y = X + C - q p as registers, then carry to zero
    pl = BF.limbs_of(p, n, k)
    R = max(m, Lq + k - 1)
    W = max(B + 2, 2 * n + k.bit_length() + 1) + 1                # |Y_i| < 2^W
    cb = W - n + 2                                                # |carry| < 2^(cb - 1)
    carries = c.signal("carry", R - 1)
    prev = c.const(0)
    for i in range(R):
        y = (inp[i] + C[i]) if i < m else c.const(0)
        for j in range(max(0, i - k + 1), min(Lq, i + 1)):
            y = y - q[j] * pl[i - j]
        if i < R - 1:
            c.hint(carries[i], (y + prev + (1 << (W + 1))) // (1 << n) - (1 << (W + 1 - n)))
            c.enforce(y + prev, carries[i] * (1 << n))
            rc = c.component("carry_range", Num2Bits(cb), i)
            c.set(rc["in"], carries[i] + (1 << (cb - 1)))
            prev = carries[i]
        else:
            c.enforce(y + prev, 0)


@template
def BigLessThanConst(c, n, k, bound):
    """in[k] (n-bit limbs) < bound, as a constraint (CheckInRangeSecp256k1's role: coordinates are reduced)"""
    inp = c.input("in", k)
    bl = BF.limbs_of(bound, n, k)
    # lexicographic comparison from the top limb: lt_i, eq_i per limb
    res = c.const(0)                                              # running "less than" of the limbs below
    for i in range(k):
        lt = c.component("lt", LessThan(n), i)
        c.set(lt["in"][0], inp[i])
        c.set(lt["in"][1], bl[i])
        eq = c.component("eq", IsEqual(), i)
        c.set(eq["in"][0], inp[i])
        c.set(eq["in"][1], bl[i])
        acc = c.signal("acc%d" % i)
        c.set(acc, lt["out"] + eq["out"] * res)                   # this limb smaller, or equal and the lower limbs smaller
        res = acc
    c.enforce(res, 1)


def _range_limbs(c, name, sig, n, k):
    for i in range(k):
        rc = c.component(name, Num2Bits(n), i)
        c.set(rc["in"], sig[i])


def _mod_check(c, name, regs, n, k, B, p):
    chk = c.component(name, CheckZeroModP(n, k, len(regs), B, p))
    for i, r in enumerate(regs):
        c.set(chk["in"][i], r)


# ---- curve operations -------------------------------------------------------------------------------------------------------------
def _curve_relations(c, n, k, p, lam, x1, y1, xo, yo, x3, y3, first_regs, B):
    """the two relations every chord / tangent step shares: lambda^2 = x1 + xo + x3, lambda (x1 - x3) = y1 + y3 (mod p)"""
    _mod_check(c, "slope", first_regs, n, k, B, p)
    sq = c.component("lam_sq", BigMultNoCarry(n, k, k))
    for i in range(k):
        c.set(sq["a"][i], lam[i])
        c.set(sq["b"][i], lam[i])
    _mod_check(c, "x_rel", [sq["out"][i] - ((x1[i] + xo[i] + x3[i]) if i < k else 0) for i in range(2 * k - 1)], n, k, B, p)
    ly = c.component("lam_dx3", BigMultNoCarry(n, k, k))
    for i in range(k):
        c.set(ly["a"][i], lam[i])
        c.set(ly["b"][i], x1[i] - x3[i])
    _mod_check(c, "y_rel", [ly["out"][i] - ((y1[i] + y3[i]) if i < k else 0) for i in range(2 * k - 1)], n, k, B, p)


@template
def EcAddUnequal(c, n, k, cv):
    """out = a + b for points with different x coordinates (secp256k1.circom Secp256k1AddUnequal)"""
    p = cv[0]
    a = c.input("a", 2, k)
    b = c.input("b", 2, k)
    out = c.output("out", 2, k)
    lam = c.signal("lambda", k)
    fn = c.function("ec_addunequal_%d_%d_%x" % (n, k, p & 0xFFFF), 4 * k, BF.build_ec_add_unequal(n, k, p), native=("ec_add", n, k, p))
    res = c.call(fn, [a[0][i] for i in range(k)] + [a[1][i] for i in range(k)] + [b[0][i] for i in range(k)] + [b[1][i] for i in range(k)])
    for i in range(k):
        c.hint(lam[i], res[i])
        c.hint(out[0][i], res[k + i])
        c.hint(out[1][i], res[2 * k + i])
    _range_limbs(c, "lam_range", lam, n, k)
    _range_limbs(c, "x_range", out[0], n, k)
    _range_limbs(c, "y_range", out[1], n, k)
    for j, nm in ((0, "x_lt_p"), (1, "y_lt_p")):
        lt = c.component(nm, BigLessThanConst(n, k, p))
        for i in range(k):
            c.set(lt["in"][i], out[j][i])
    B = 2 * n + k.bit_length() + 2
    sl = c.component("lam_dx", BigMultNoCarry(n, k, k))
    for i in range(k):
        c.set(sl["a"][i], lam[i])
        c.set(sl["b"][i], b[0][i] - a[0][i])
    first = [sl["out"][i] - ((b[1][i] - a[1][i]) if i < k else 0) for i in range(2 * k - 1)]
    _curve_relations(c, n, k, p, lam, a[0], a[1], b[0], b[1], out[0], out[1], first, B)


@template
def EcDouble(c, n, k, cv):
    """out = 2 a (secp256k1.circom Secp256k1Double): lambda (2 y1) = 3 x1^2"""
    p = cv[0]
    a = c.input("in", 2, k)
    out = c.output("out", 2, k)
    lam = c.signal("lambda", k)
    fn = c.function("ec_double_%d_%d_%x" % (n, k, p & 0xFFFF), 2 * k, BF.build_ec_double(n, k, p), native=("ec_double", n, k, p))
    res = c.call(fn, [a[0][i] for i in range(k)] + [a[1][i] for i in range(k)])
    for i in range(k):
        c.hint(lam[i], res[i])
        c.hint(out[0][i], res[k + i])
        c.hint(out[1][i], res[2 * k + i])
    _range_limbs(c, "lam_range", lam, n, k)
    _range_limbs(c, "x_range", out[0], n, k)
    _range_limbs(c, "y_range", out[1], n, k)
    for j, nm in ((0, "x_lt_p"), (1, "y_lt_p")):
        lt = c.component(nm, BigLessThanConst(n, k, p))
        for i in range(k):
            c.set(lt["in"][i], out[j][i])
    B = 2 * n + k.bit_length() + 4
    sl = c.component("lam_2y", BigMultNoCarry(n, k, k))
    xx = c.component("x_sq", BigMultNoCarry(n, k, k))
    for i in range(k):
        c.set(sl["a"][i], lam[i])
        c.set(sl["b"][i], a[1][i] * 2)
        c.set(xx["a"][i], a[0][i])
        c.set(xx["b"][i], a[0][i])
    first = [sl["out"][i] - xx["out"][i] * 3 for i in range(2 * k - 1)]
    _curve_relations(c, n, k, p, lam, a[0], a[1], a[0], a[1], out[0], out[1], first, B)


@template
def EcScalarMult(c, n, k, cv):
    """out = scalar * point, MSB-first double-and-add over the n k bits of the scalar (secp256k1.circom Secp256k1ScalarMult):
    until the first 1 bit the partial result is the point itself"""
    scalar = c.input("scalar", k)
    point = c.input("point", 2, k)
    out = c.output("out", 2, k)
    bits = []
    for i in range(k):
        nb = c.component("n2b", Num2Bits(n), i)
        c.set(nb["in"], scalar[i])
        bits += [nb["out"][j] for j in range(n)]
    nbits = n * k
    cur = [[point[0][i] for i in range(k)], [point[1][i] for i in range(k)]]
    has_prev = bits[nbits - 1]
    for t in range(nbits - 2, -1, -1):
        dbl = c.component("doubler", EcDouble(n, k, cv), t)
        for j in range(2):
            for i in range(k):
                c.set(dbl["in"][j][i], cur[j][i])
        add = c.component("adder", EcAddUnequal(n, k, cv), t)
        for j in range(2):
            for i in range(k):
                c.set(add["a"][j][i], dbl["out"][j][i])
                c.set(add["b"][j][i], point[j][i])
        bit = bits[t]
        nxt = [[None] * k, [None] * k]
        for j in range(2):
            for i in range(k):
                # has_prev ? (bit ? add : dbl) : point
                sel = c.signal("sel_%d_%d_%d" % (t, j, i))
                c.set(sel, dbl["out"][j][i] + bit * (add["out"][j][i] - dbl["out"][j][i]))
                r = c.signal("part_%d_%d_%d" % (t, j, i))
                c.set(r, point[j][i] + has_prev * (sel - point[j][i]))
                nxt[j][i] = r
        hp = c.signal("has_prev_%d" % t)
        c.set(hp, has_prev + bit - has_prev * bit)
        has_prev = hp
        cur = nxt
    for j in range(2):
        for i in range(k):
            c.set(out[j][i], cur[j][i])


def fixed_base_table(cv, n, k, stride):
    """T[w][d] = d * 2^(stride w) * G for d = 1 .. 2^stride - 1 (ecdsa.circom get_g_pow_stride8_table); d = 0 -> a dummy
    point ((2^stride + 1) times the window's base: never equal to a table entry or to an accumulated sum of lower windows, so
    the adder that is computed and discarded for a zero digit stays satisfiable)"""
    G = (cv[2], cv[3])
    nw = -(-(n * k) // stride)
    T = []
    base = G
    for w in range(nw):
        row = [None]
        acc = None
        for d in range(1, 1 << stride):
            acc = ec_add(cv, acc, base)
            row.append(acc)
        nxt = ec_add(cv, acc, base)                                # 2^stride * base: the next window's base
        row[0] = ec_add(cv, nxt, base)                             # (2^stride + 1) * base: no digit of any window selects it
        T.append(row)
        base = nxt
    return T


@template
def EcFixedBaseMult(c, n, k, cv, stride):
    """out = scalar * G with a precomputed table of the multiples of G, `stride` bits at a time (ecdsa.circom ECDSAPrivToPub):
    per window an indicator vector of the digit selects the table entry (constants), windows with digit 0 are skipped"""
    scalar = c.input("scalar", k)
    out = c.output("out", 2, k)
    T = fixed_base_table(cv, n, k, stride)
    bits = []
    for i in range(k):
        nb = c.component("n2b", Num2Bits(n), i)
        c.set(nb["in"], scalar[i])
        bits += [nb["out"][j] for j in range(n)]
    nw = len(T)
    cur = None
    has_prev = None
    for w in range(nw):
        wb = bits[w * stride:(w + 1) * stride]
        nd = 1 << len(wb)
        # indicator of the digit: ind[d] = prod over bits (bit or 1 - bit), built as a binary tree of products
        level = [c.const(1)]
        for bi, b_ in enumerate(wb):
            nl = []
            for d, e in enumerate(level):
                hi = c.signal("ind_%d_%d_%d" % (w, bi, d))
                c.set(hi, e * b_)
                nl.append((e - hi, hi))
            level = [lo for lo, _ in nl] + [hi for _, hi in nl]    # digit d + (bit << bi): the upper half has the bit set
        ind = level
        sel = [[None] * k, [None] * k]
        for j in range(2):
            for i in range(k):
                acc = c.const(0)
                for d in range(nd):
                    acc = acc + ind[d] * BF.limbs_of(T[w][d][j], n, k)[i]
                s = c.signal("tab_%d_%d_%d" % (w, j, i))
                c.set(s, acc)
                sel[j][i] = s
        nz = c.signal("nz_%d" % w)
        c.set(nz, 1 - ind[0])
        if cur is None:
            cur, has_prev = sel, nz
            continue
        add = c.component("adder", EcAddUnequal(n, k, cv), w)
        for j in range(2):
            for i in range(k):
                c.set(add["a"][j][i], cur[j][i])
                c.set(add["b"][j][i], sel[j][i])
        nxt = [[None] * k, [None] * k]
        both = c.signal("both_%d" % w)
        c.set(both, has_prev * nz)
        for j in range(2):
            for i in range(k):
                # has_prev & nz -> sum; has_prev & !nz -> cur; !has_prev -> sel (the table entry, or the dummy)
                t1 = c.signal("keep_%d_%d_%d" % (w, j, i))
                c.set(t1, sel[j][i] + has_prev * (cur[j][i] - sel[j][i]))
                r = c.signal("acc_%d_%d_%d" % (w, j, i))
                c.set(r, t1 + both * (add["out"][j][i] - t1))
                nxt[j][i] = r
        hp = c.signal("has_prev_%d" % w)
        c.set(hp, has_prev + nz - both)
        has_prev = hp
        cur = nxt
    for j in range(2):
        for i in range(k):
            c.set(out[j][i], cur[j][i])


@template
def BigMultModPConst(c, n, k, p):
    """out = a b mod p, p a constant (bigint.circom BigMultModP with the modulus folded in)"""
    a = c.input("a", k)
    b = c.input("b", k)
    out = c.output("out", k)
    fn = c.function("prod_mod_%d_%d_%x" % (n, k, p & 0xFFFF), 2 * k, _build_prod_mod(n, k, p))
    res = c.call(fn, [a[i] for i in range(k)] + [b[i] for i in range(k)])
    for i in range(k):
        c.hint(out[i], res[i])
    _range_limbs(c, "out_range", out, n, k)
    lt = c.component("out_lt_p", BigLessThanConst(n, k, p))
    for i in range(k):
        c.set(lt["in"][i], out[i])
    ab = c.component("ab", BigMultNoCarry(n, k, k))
    for i in range(k):
        c.set(ab["a"][i], a[i])
        c.set(ab["b"][i], b[i])
    _mod_check(c, "rel", [ab["out"][i] - (out[i] if i < k else 0) for i in range(2 * k - 1)], n, k, 2 * n + k.bit_length() + 1, p)


def _build_prod_mod(n, k, p_int):
    def build(f, *args):
        p = [f.var(v) for v in BF.limbs_of(p_int, n, k)]
        return BF.prod_mod(f, n, k, list(args[:k]), list(args[k:2 * k]), p)
    return build


@template
def BigModInvConst(c, n, k, p):
    """out = in^-1 mod p (bigint.circom BigModInv): hint by the run-time function mod_inv, checked by in * out = 1 (mod p)"""
    inp = c.input("in", k)
    out = c.output("out", k)
    fn = c.function("mod_inv_%d_%d_%x" % (n, k, p & 0xFFFF), k, BF.build_mod_inv(n, k, p), native=("mod_inv", n, k, p))
    res = c.call(fn, [inp[i] for i in range(k)])
    for i in range(k):
        c.hint(out[i], res[i])
    _range_limbs(c, "out_range", out, n, k)
    ab = c.component("ab", BigMultNoCarry(n, k, k))
    for i in range(k):
        c.set(ab["a"][i], inp[i])
        c.set(ab["b"][i], out[i])
    _mod_check(c, "rel", [ab["out"][i] - (1 if i == 0 else 0) for i in range(2 * k - 1)], n, k, 2 * n + k.bit_length() + 1, p)


@template
def ECDSAVerifyNoPubkeyCheck(c, n, k, cv, stride):
    """result = 1 iff (r, s) is a valid signature of msghash under pubkey (ecdsa.circom ECDSAVerifyNoPubkeyCheck):
    s^-1 mod N, u1 = s^-1 h, u2 = s^-1 r, u1 G + u2 Q, x coordinate against r.  The pubkey is NOT checked to be on the curve
    (hence the name); the corner x >= N (probability 2^-128 for secp256k1) is not reduced."""
    p, order = cv[0], cv[1]
    r = c.input("r", k)
    s = c.input("s", k)
    msghash = c.input("msghash", k)
    pubkey = c.input("pubkey", 2, k)
    result = c.output("result")
    sinv = c.component("sinv", BigModInvConst(n, k, order))
    for i in range(k):
        c.set(sinv["in"][i], s[i])
    gc = c.component("g_coeff", BigMultModPConst(n, k, order))
    pc = c.component("pubkey_coeff", BigMultModPConst(n, k, order))
    for i in range(k):
        c.set(gc["a"][i], sinv["out"][i])
        c.set(gc["b"][i], msghash[i])
        c.set(pc["a"][i], sinv["out"][i])
        c.set(pc["b"][i], r[i])
    gm = c.component("g_mult", EcFixedBaseMult(n, k, cv, stride))
    pm = c.component("pubkey_mult", EcScalarMult(n, k, cv))
    for i in range(k):
        c.set(gm["scalar"][i], gc["out"][i])
        c.set(pm["scalar"][i], pc["out"][i])
    for j in range(2):
        for i in range(k):
            c.set(pm["point"][j][i], pubkey[j][i])
    sm = c.component("sum", EcAddUnequal(n, k, cv))
    for j in range(2):
        for i in range(k):
            c.set(sm["a"][j][i], gm["out"][j][i])
            c.set(sm["b"][j][i], pm["out"][j][i])
    acc = c.const(1)
    for i in range(k):
        eq = c.component("x_eq_r", IsEqual(), i)
        c.set(eq["in"][0], sm["out"][0][i])
        c.set(eq["in"][1], r[i])
        if i == 0:
            acc = eq["out"]
        else:
            a2 = c.signal("all_eq_%d" % i)
            c.set(a2, acc * eq["out"])
            acc = a2
    c.set(result, acc)


# ---- host-side synthesis of valid inputs -------------------------------------------------------------------------------------------
def sign(cv, n, k, rnd):
    """a random key pair, message hash and signature: the main inputs of ECDSAVerifyNoPubkeyCheck in declaration order
    (r, s, msghash, pubkey[0], pubkey[1]), as limb lists"""
    p, order = cv[0], cv[1]
    G = (cv[2], cv[3])
    while True:
        d = rnd.randrange(1, order)
        Q = ec_mul(cv, d, G)
        h = rnd.randrange(order)
        kk = rnd.randrange(1, order)
        R = ec_mul(cv, kk, G)
        r = R[0] % order
        if r == 0 or R[0] >= order:
            continue
        s = pow(kk, order - 2, order) * (h + r * d) % order
        if s == 0:
            continue
        return BF.limbs_of(r, n, k) + BF.limbs_of(s, n, k) + BF.limbs_of(h, n, k) + BF.limbs_of(Q[0], n, k) + BF.limbs_of(Q[1], n, k)
