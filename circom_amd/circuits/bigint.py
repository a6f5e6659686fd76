"""Non-native big-integer arithmetic in the shape of 0xPARC circom-ecdsa's `bigint.circom` / `bigint_func.circom`
(BASELINE.json config 5 is built from these): k limbs of n bits, witness values computed by circom FUNCTIONS whose
control flow depends on run-time values (`long_div` -> `short_div` with its correction branches, `long_gt`,
`long_sub`, `long_scalar_mult`), results range-checked and tied to the inputs by constraints.

circom-ecdsa itself is absent from the reference tree, so the functions are re-authored from the algorithm (Knuth D
with the normalisation trick of bigint_func.circom: scale = 2^n \\ (1 + b[k-1])); they run as tier-2 bytecode
(frontend/rtcode.py).  The constraint side is the compact variant that fits one field element (n*2k <= 250 bits):
`div * b + mod === a` on the packed values plus limb range checks and `mod < b` — enough to pin every witness value;
the limb-wise carry checks of circom-ecdsa (CheckCarryToZero) are not reproduced.
"""
from ..frontend.dsl import template
from .basic import Num2Bits
from .stdlib import LessThan


# ---- functions (bigint_func.circom) -------------------------------------------------------------------------------
def _long_gt(f, n, k, a, b):
    """a > b on k limbs (most significant limb decides): result register 0/1"""
    res = f.var(0)
    decided = f.var(0)
    for i in range(k - 1, -1, -1):
        with f.if_(decided.eq(0)):
            with f.if_(a[i].gt(b[i])):
                res.set(1)
                decided.set(1)
            with f.if_(a[i].lt(b[i])):
                decided.set(1)
    return res


def _long_sub(f, n, k, a, b):
    """a - b on k limbs, a >= b"""
    out = [None] * k
    borrow = f.var(0)
    for i in range(k):
        t = f.var(b[i] + borrow)
        d = f.var(0)
        with f.if_(a[i].geq(t)):
            d.set(a[i] - t)
            borrow.set(0)
        with f.else_():
            d.set(a[i] + (1 << n) - t)
            borrow.set(1)
        out[i] = d
    return out


def _long_scalar_mult(f, n, k, s, a):
    """s * a: k+1 limbs"""
    out = []
    carry = f.var(0)
    for i in range(k):
        t = f.var(s * a[i] + carry)
        out.append(f.var(t % (1 << n)))
        carry.set(t // (1 << n))
    out.append(f.var(carry + 0))
    return out


def _short_div(f, n, k, a, b):
    """quotient digit of a (k+1 limbs) by b (k limbs), a < 2^n * b   (bigint_func.circom short_div)"""
    scale = f.var(f.lift(1 << n) // (b[k - 1] + 1))
    norm_a = _long_scalar_mult(f, n, k + 1, scale, a)          # k+2 limbs
    norm_b = _long_scalar_mult(f, n, k, scale, b)              # k+1 limbs
    qhat = f.var(0)
    with f.if_(norm_a[k].neq(0)):
        qhat.set((norm_a[k] * (1 << n) + norm_a[k - 1]) // norm_b[k - 1])
    with f.else_():
        qhat.set(norm_a[k - 1] // norm_b[k - 1])
    with f.if_(qhat.gt((1 << n) - 1)):
        qhat.set((1 << n) - 1)
    mult = _long_scalar_mult(f, n, k, qhat, b)                 # k+1 limbs
    out = f.var(qhat + 0)
    with f.if_(_long_gt(f, n, k + 1, mult, a)):
        mult1 = _long_sub(f, n, k + 1, mult, list(b) + [f.var(0)])
        with f.if_(_long_gt(f, n, k + 1, mult1, a)):
            out.set(qhat - 2)
        with f.else_():
            out.set(qhat - 1)
    return out


def build_long_div(n, k):
    """long_div(n, k, k, a[2k], b[k]) -> div[k+1], mod[k]"""
    def build(f, *args):
        a = [args[i] for i in range(2 * k)]
        b = [args[2 * k + i] for i in range(k)]
        rem = [f.var(a[i] + 0) for i in range(2 * k)] + [f.var(0)]
        div = [None] * (k + 1)
        for i in range(k, -1, -1):
            if i == k:
                dividend = rem[k:2 * k] + [f.var(0)]
            else:
                dividend = rem[i:i + k + 1]
            d = _short_div(f, n, k, dividend, b)
            div[i] = d
            mult = _long_scalar_mult(f, n, k, d, b)            # k+1 limbs, to be subtracted at limb offset i
            sub = _long_sub(f, n, k + 1, rem[i:i + k + 1], mult)
            for j in range(k + 1):
                rem[i + j] = sub[j]
        return div + rem[:k]
    return build


# ---- templates ----------------------------------------------------------------------------------------------------
@template
def BigMod(c, n, k):
    """a[2k] = div[k+1] * b[k] + mod[k], mod < b   (bigint.circom BigMod, compact constraint side)"""
    assert n * 2 * k <= 250
    a = c.input("a", 2 * k)
    b = c.input("b", k)
    div = c.output("div", k + 1)
    mod = c.output("mod", k)
    fn = c.function("long_div_%d_%d" % (n, k), 3 * k, build_long_div(n, k))
    res = c.call(fn, [a[i] for i in range(2 * k)] + [b[i] for i in range(k)])
    for i in range(k + 1):
        c.hint(div[i], res[i])
    for i in range(k):
        c.hint(mod[i], res[k + 1 + i])
    # limbs are n-bit numbers
    for i in range(k + 1):
        rc = c.component("div_range", Num2Bits(n), i)
        c.set(rc["in"], div[i])
    for i in range(k):
        rc = c.component("mod_range", Num2Bits(n), i)
        c.set(rc["in"], mod[i])
    A = sum((a[i] * (1 << (n * i)) for i in range(1, 2 * k)), a[0] + 0)
    Bv = sum((b[i] * (1 << (n * i)) for i in range(1, k)), b[0] + 0)
    D = sum((div[i] * (1 << (n * i)) for i in range(1, k + 1)), div[0] + 0)
    M = sum((mod[i] * (1 << (n * i)) for i in range(1, k)), mod[0] + 0)
    c.enforce(D * Bv + M, A)
    lt = c.component("lt", LessThan(n * k))
    c.set(lt["in"][0], M)
    c.set(lt["in"][1], Bv)
    c.enforce(lt["out"], 1)


@template
def BigMultModP(c, n, k):
    """out = a * b mod p on k-limb numbers (bigint.circom BigMultModP): the field multiplication of a foreign curve"""
    assert n * 2 * k <= 250
    a = c.input("a", k)
    b = c.input("b", k)
    p = c.input("p", k)
    out = c.output("out", k)
    A = sum((a[i] * (1 << (n * i)) for i in range(1, k)), a[0] + 0)
    Bv = sum((b[i] * (1 << (n * i)) for i in range(1, k)), b[0] + 0)
    prod = c.signal("prod")
    c.set(prod, A * Bv)
    bits = c.component("prod_bits", Num2Bits(2 * n * k))
    c.set(bits["in"], prod)
    bm = c.component("big_mod", BigMod(n, k))
    for i in range(2 * k):
        limb = sum((bits["out"][n * i + j] * (1 << j) for j in range(1, n)), bits["out"][n * i] + 0)
        c.set(bm["a"][i], limb)
    for i in range(k):
        c.set(bm["b"][i], p[i])
    for i in range(k):
        c.set(out[i], bm["mod"][i])
