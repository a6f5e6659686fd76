"""circom FUNCTIONS of 0xPARC circom-ecdsa's `bigint_func.circom` / `secp256k1_func.circom`, re-authored as tier-2 bytecode
(frontend/rtcode.py): long arithmetic on k limbs of n bits whose control flow depends on run-time values — the `<--` side of
BASELINE config 5 (secp256k1 ECDSA verification inside a circuit over the BLS12-381 scalar field).

circom-ecdsa is not in the reference tree; the functions follow its published algorithms: `prod` (schoolbook with deferred
carries), `long_div` / `short_div` (Knuth D, normalised by 2^n \\ (1 + b[k-1])), `mod_exp` (square-and-multiply over the bits
of the exponent, a RUN-TIME loop with a run-time indexed limb), `mod_inv` = `mod_exp(a, p - 2)`, the chord / tangent formulas
of `secp256k1_addunequal_func` / `secp256k1_double_func`.

What runs them:
  * the REFERENCE RUNTIME, through oracle/emit_ref_cpp.py (labels + gotos over the reference's own `Fr_*` calls): the golden
    `.wtns` files of the ECDSA circuit come from there;
  * the oracle's interpreter (oracle/tape_eval.py) — ~10^6 steps per modular inverse, so the three functions that contain one
    carry a NATIVE tag (`fn.native = (kind, n, k, modulus)`): a pure function of its arguments has ONE right answer, and both
    the oracle (Python integers) and the device (csrc/cw_kernels.hip: binary-GCD inverse + Montgomery products modulo the
    foreign prime) may compute it directly.  tests/test_ecdsa.py pins native == bytecode on random and edge arguments, on the
    CPU against the interpreter and against the reference runtime executing the emitted body.
"""
from .bigint import _long_gt, _long_sub, _long_scalar_mult, _short_div


def limbs_of(x: int, n: int, k: int):
    return [(x >> (n * i)) & ((1 << n) - 1) for i in range(k)]


def int_of(limbs, n: int) -> int:
    return sum(int(v) << (n * i) for i, v in enumerate(limbs))


# ---- builders on lists of registers (unrolled over the limbs at build time) -----------------------------------------------
def long_add(f, n, k, a, b):
    """a + b: k + 1 limbs"""
    out = []
    carry = f.var(0)
    for i in range(k):
        t = f.var(a[i] + b[i] + carry)
        out.append(f.var(t % (1 << n)))
        carry.set(t // (1 << n))
    out.append(f.var(carry + 0))
    return out


def prod(f, n, k, a, b):
    """a * b: 2k limbs (column sums first: k products of 2n bits stay far below the field's 250+ bits, then one carry pass)"""
    out = []
    carry = f.var(0)
    for i in range(2 * k - 1):
        col = None
        for j in range(max(0, i - k + 1), min(k, i + 1)):
            t = a[j] * b[i - j]
            col = t if col is None else col + t
        t = f.var(col + carry)
        out.append(f.var(t % (1 << n)))
        carry.set(t // (1 << n))
    out.append(f.var(carry + 0))
    return out


def long_div(f, n, k, m, a, b):
    """a (k + m limbs) = div (m + 1 limbs) * b (k limbs) + mod (k limbs); b[k-1] != 0   (bigint_func.circom long_div)"""
    rem = [f.var(a[i] + 0) for i in range(k + m)] + [f.var(0)]
    div = [None] * (m + 1)
    for i in range(m, -1, -1):
        if i == m:
            dividend = rem[m:m + k] + [f.var(0)]
        else:
            dividend = rem[i:i + k + 1]
        d = _short_div(f, n, k, dividend, b)
        div[i] = d
        mult = _long_scalar_mult(f, n, k, d, b)
        sub = _long_sub(f, n, k + 1, rem[i:i + k + 1], mult)
        for j in range(k + 1):
            rem[i + j] = sub[j]
    return div, rem[:k]


def prod_mod(f, n, k, a, b, p):
    return long_div(f, n, k, k, prod(f, n, k, a, b), p)[1]


def sub_mod(f, n, k, a, b, p):
    """a - b mod p for a, b < p"""
    out = [f.var(0) for _ in range(k)]
    with f.if_(_long_gt(f, n, k, b, a)):
        t = long_add(f, n, k, a, p)                               # a + p - b: k + 1 limbs, the top one ends as 0
        d = _long_sub(f, n, k + 1, t, list(b) + [f.var(0)])
        for i in range(k):
            out[i].set(d[i])
    with f.else_():
        d = _long_sub(f, n, k, a, b)
        for i in range(k):
            out[i].set(d[i])
    return out


def add_mod(f, n, k, a, b, p):
    s = long_add(f, n, k, a, b)
    out = [f.var(s[i] + 0) for i in range(k)]
    pp = list(p) + [f.var(0)]
    with f.if_(_long_gt(f, n, k + 1, pp, s).eq(0)):               # s >= p
        d = _long_sub(f, n, k + 1, s, pp)
        for i in range(k):
            out[i].set(d[i])
    return out


def mod_exp(f, n, k, a, p, e):
    """a^e mod p: square-and-multiply from the top bit of e down — a run-time loop of n*k trips whose body reads the limb
    e[i \\ n] through a run-time index (bigint_func.circom mod_exp)"""
    earr = f.array(k, init=e)
    out = f.array(k, init=[1] + [0] * (k - 1))
    i = f.var(n * k)
    with f.loop() as L:
        L.break_unless(i.neq(0))
        i.set(i - 1)
        cur = [out[j] for j in range(k)]
        sq = prod_mod(f, n, k, cur, cur, p)
        for j in range(k):
            out[j].set(sq[j])
        limb = earr.load(i // n)
        with f.if_((limb >> (i % n)) & 1):
            ml = prod_mod(f, n, k, [out[j] for j in range(k)], a, p)
            for j in range(k):
                out[j].set(ml[j])
    return [out[j] for j in range(k)]


def mod_inv(f, n, k, a, p):
    """a^(p-2) mod p (p prime); 0 for a = 0   (bigint_func.circom mod_inv)"""
    two = [f.var(2)] + [f.var(0) for _ in range(k - 1)]
    return mod_exp(f, n, k, a, p, _long_sub(f, n, k, p, two))


# ---- whole functions (what a template calls) ----------------------------------------------------------------------------------
def native_n_args(kind: str, k: int, p: int) -> int:
    """arguments a native form reads (its results follow them); for "long_div" the tag's last field is m"""
    return {"mod_inv": k, "ec_add": 4 * k, "ec_double": 2 * k, "long_div": 2 * k + p}[kind]


def build_long_div(n, k, m):
    """long_div(n, k, m, a[k + m], b[k]) -> div[m + 1] ++ mod[k]"""
    def build(f, *args):
        div, mod = long_div(f, n, k, m, list(args[:k + m]), list(args[k + m:k + m + k]))
        return div + mod
    return build


def build_mod_inv(n, k, p_int):
    """mod_inv(n, k, a[k], p) with the prime as a constant array (circom-ecdsa passes get_secp256k1_prime / _order)"""
    def build(f, *args):
        p = [f.var(v) for v in limbs_of(p_int, n, k)]
        return mod_inv(f, n, k, list(args[:k]), p)
    return build


def build_ec_add_unequal(n, k, p_int):
    """secp256k1_addunequal_func(n, k, x1, y1, x2, y2) -> x3 ++ y3: lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2,
    y3 = lambda (x1 - x3) - y1, all modulo the constant prime; also returns lambda (the circuit constrains through it)"""
    def build(f, *args):
        x1, y1, x2, y2 = (list(args[j * k:(j + 1) * k]) for j in range(4))
        p = [f.var(v) for v in limbs_of(p_int, n, k)]
        dx = sub_mod(f, n, k, x2, x1, p)
        dy = sub_mod(f, n, k, y2, y1, p)
        lam = prod_mod(f, n, k, dy, mod_inv(f, n, k, dx, p), p)
        l2 = prod_mod(f, n, k, lam, lam, p)
        x3 = sub_mod(f, n, k, sub_mod(f, n, k, l2, x1, p), x2, p)
        y3 = sub_mod(f, n, k, prod_mod(f, n, k, lam, sub_mod(f, n, k, x1, x3, p), p), y1, p)
        return lam + x3 + y3
    return build


def build_ec_double(n, k, p_int):
    """secp256k1_double_func(n, k, x1, y1) -> lambda ++ x3 ++ y3: lambda = 3 x1^2 / (2 y1)   (curve coefficient a = 0)"""
    def build(f, *args):
        x1, y1 = list(args[:k]), list(args[k:2 * k])
        p = [f.var(v) for v in limbs_of(p_int, n, k)]
        xx = prod_mod(f, n, k, x1, x1, p)
        num = add_mod(f, n, k, add_mod(f, n, k, xx, xx, p), xx, p)
        den = add_mod(f, n, k, y1, y1, p)
        lam = prod_mod(f, n, k, num, mod_inv(f, n, k, den, p), p)
        l2 = prod_mod(f, n, k, lam, lam, p)
        x3 = sub_mod(f, n, k, sub_mod(f, n, k, l2, x1, p), x1, p)
        y3 = sub_mod(f, n, k, prod_mod(f, n, k, lam, sub_mod(f, n, k, x1, x3, p), p), y1, p)
        return lam + x3 + y3
    return build


# ---- the same functions on Python integers (the NATIVE semantics: oracle shortcut, host-side input synthesis) ---------------
def native_eval(kind: str, n: int, k: int, p: int, args):
    """args / results: lists of limb values (ints).  Must equal the bytecode on every argument the templates can pass."""
    if kind == "mod_inv":
        a = int_of(args[:k], n)
        return limbs_of(pow(a, p - 2, p), n, k)
    if kind == "ec_add":
        x1, y1, x2, y2 = (int_of(args[j * k:(j + 1) * k], n) for j in range(4))
        lam = (y2 - y1) * pow((x2 - x1) % p, p - 2, p) % p
        x3 = (lam * lam - x1 - x2) % p
        y3 = (lam * (x1 - x3) - y1) % p
        return limbs_of(lam, n, k) + limbs_of(x3, n, k) + limbs_of(y3, n, k)
    if kind == "ec_double":
        x1, y1 = int_of(args[:k], n), int_of(args[k:2 * k], n)
        lam = 3 * x1 * x1 * pow(2 * y1 % p, p - 2, p) % p
        x3 = (lam * lam - 2 * x1) % p
        y3 = (lam * (x1 - x3) - y1) % p
        return limbs_of(lam, n, k) + limbs_of(x3, n, k) + limbs_of(y3, n, k)
    if kind == "long_div":
        # the fourth field of the tag is m, not a modulus: a[k + m], b[k] -> div[m + 1] ++ mod[k].  Contract: proper limbs and
        # b[k-1] != 0 (then div fits m + 1 limbs and the pair is unique, so Knuth D in base 2^n - the body - finds the same one);
        # templates only tag calls whose divisor is a compile-time constant (CheckZeroModP divides by the foreign prime)
        m = p
        a, b = int_of(args[:k + m], n), int_of(args[k + m:2 * k + m], n)
        if args[2 * k + m - 1] == 0:
            raise ZeroDivisionError("long_div native form: the divisor's top limb is zero (contract of the tag)")
        return limbs_of(a // b, n, m + 1) + limbs_of(a % b, n, k)
    raise ValueError(kind)
