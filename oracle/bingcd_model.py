"""Limb-level model of the device's modular inversion (csrc/fp256.hip.h: fe_inv) — TEST INFRASTRUCTURE.

The reference inverts with GMP (`Fr_inv` -> `mpz_invert`, generic/fr.cpp:2895-2906; inverse of 0 is 0).  The
device cannot branch per lane, so it uses a constant-time binary extended GCD in the style of Pornin,
"Optimized Binary GCD for Modular Inversion" (2020): K = 30 halving steps at a time are run on 64-bit
approximations of (a, b) (low 30 bits exact, top 34 bits of the longer of the two), producing update factors
|f|,|g| <= 2^30, which are then applied to the full-width values:
      a, b <- |f0 a + g0 b| / 2^30, |f1 a + g1 b| / 2^30          (exact divisions)
      u, v <- +-(f0 u + g0 v) / 2^30 mod m, +-(f1 u + g1 v) / 2^30 mod m
with  a = u*y, b = v*y (mod m) invariant.  After ceil((2*len(m) - 1) / 30) rounds b = gcd = 1 and v = 1/y.
This file restates the device code with Python integers held in the same limb discipline (unsigned 32-bit
limbs, offset factors f' = f + 2^30, wrap-around 288-bit intermediates) so that tests can check the
arithmetic identities the kernel relies on; tests compare it with pow(y, -1, m)."""
from __future__ import annotations

K = 30
M64 = (1 << 64) - 1
M288 = (1 << 288) - 1


def approx(a: int, b: int):
    n = max(a.bit_length(), b.bit_length())
    if n <= 64:
        return a, b
    s = n - 34
    lo = (1 << K) - 1
    return (a & lo) | ((a >> s) << K), (b & lo) | ((b >> s) << K)


def inner(xa: int, xb: int):
    f0, g0, f1, g1 = 1, 0, 0, 1
    for _ in range(K):
        if xa & 1:
            if xa < xb:
                xa, xb, f0, f1, g0, g1 = xb, xa, f1, f0, g1, g0
            xa -= xb
            f0 -= f1
            g0 -= g1
        xa >>= 1
        f1 <<= 1
        g1 <<= 1
    return f0, g0, f1, g1


def lincomb_wrap(x: int, y: int, f: int, g: int) -> int:
    """f*x + g*y as the device computes it: offset factors, 288-bit wrap-around two's complement."""
    fp, gp = f + (1 << K), g + (1 << K)
    assert 0 <= fp <= 1 << 31 and 0 <= gp <= 1 << 31
    U = (fp * x + gp * y) & M288
    W = ((x + y) << K) & M288
    return (U - W) & M288


def inv_mod(y: int, m: int) -> int:
    a, b, u, v = y % m, m, 1, 0
    ninv = (-pow(m, -1, 1 << K)) % (1 << K)
    rounds = -(-(2 * m.bit_length() - 1) // K)
    for _ in range(rounds):
        f0, g0, f1, g1 = inner(*approx(a, b))
        assert abs(f0) + abs(g0) <= 1 << K and abs(f1) + abs(g1) <= 1 << K
        new = []
        for f, g in ((f0, g0), (f1, g1)):
            d = lincomb_wrap(a, b, f, g)
            neg = d >> 287                                  # sign of the 288-bit two's complement value
            if neg:
                d = (-d) & M288
            assert d & ((1 << K) - 1) == 0 and d >> K < 1 << 256
            e = lincomb_wrap(u, v, f, g)
            if neg:
                e = (-e) & M288
            t = (e + (m << K)) & M288                       # in (0, 2^31 m): non-negative
            k = (t * ninv) & ((1 << K) - 1)
            t = t + k * m
            assert t & ((1 << K) - 1) == 0 and t < 1 << 288
            t >>= K
            assert t < 3 * m
            if t >= m:
                t -= m
            if t >= m:
                t -= m
            new.append((d >> K, t))
        (a, u), (b, v) = new
    assert a == 0 and (b == 1 or y % m == 0)
    return v if y % m else 0
