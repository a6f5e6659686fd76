"""Compile-time prime-field arithmetic of the host front-end.

Role of `circom_algebra::modular_arithmetic` (circom_algebra/src/modular_arithmetic.rs:26-215)
in the reference: constant folding while a circuit is traced, coefficient algebra of
constraints, Montgomery pre-scaling of constants for the device tape.  Values are canonical
Python ints in [0, q).  (The oracle has its own independent restatement in oracle/field.py;
tests check the two against each other and against the compiled reference library.)
"""
from __future__ import annotations

# program_structure/src/utils/constants.rs:3-13 — the primes `--prime` accepts
PRIMES = {
    "bn128": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "bls12381": 52435875175126190479447740508185965837690552500527637822603658699938581184513,
    "goldilocks": 18446744069414584321,
    "grumpkin": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "pallas": 28948022309329048855892746252171976963363056481941560715954676764349967630337,
    "vesta": 28948022309329048855892746252171976963363056481941647379679742748393362948097,
    "secq256r1": 115792089210356248762697446949407573530086143415290314195533631308867097853951,
    "bls12377": 8444461749428370424248824938781546531375899335154063827935233455917409239041,
}


class ArithmeticError_(Exception):
    """modular_arithmetic.rs:4-7 (DivisionByZero / BitOverFlowInShift)."""


class Fp:
    """Field descriptor + operator table.  One instance per prime."""

    def __init__(self, q: int, name: str = ""):
        self.q = q
        self.name = name
        self.bits = q.bit_length()
        self.mask = (1 << self.bits) - 1
        self.half = q // 2
        self.n64 = (self.bits + 63) // 64
        self.n32 = self.n64 * 2
        self.R = 1 << (64 * self.n64)
        self.Rinv = pow(self.R, -1, q)
        self.R2 = (self.R * self.R) % q
        # device Montgomery radix R' = 2^261 (9 x 29-bit limbs, csrc/fp256.hip.h); only the schedule knows it
        self.RDEV_BITS = 261
        self.Rdev = (1 << self.RDEV_BITS) % q
        self.Rdev2 = (self.Rdev * self.Rdev) % q
        # -q^-1 mod 2^32 / 2^64 (c_code_generator.rs:1093-1094 computes the 64-bit one)
        self.np64 = (-pow(q, -1, 1 << 64)) % (1 << 64)
        self.np32 = (-pow(q, -1, 1 << 32)) % (1 << 32)

    # canonical <-> Montgomery
    def to_mont(self, x):
        return (x * self.R) % self.q

    def from_mont(self, x):
        return (x * self.Rinv) % self.q

    def val(self, x):
        return x - self.q if x > self.half else x

    def _wrap(self, v):
        v &= self.mask
        return v - self.q if v >= self.q else v

    def add(self, x, y): return (x + y) % self.q
    def sub(self, x, y): return (x - y) % self.q
    def mul(self, x, y): return (x * y) % self.q
    def neg(self, x): return (-x) % self.q

    def inv(self, x):
        return 0 if x % self.q == 0 else pow(x, -1, self.q)

    def div(self, x, y):
        # run-time semantics (generic/fr.cpp:2895-2912): x/0 = 0
        return (x * self.inv(y)) % self.q

    def idiv(self, x, y):
        if y == 0:
            raise ArithmeticError_("DivisionByZero")
        return x // y

    def mod(self, x, y):
        if y == 0:
            raise ArithmeticError_("DivisionByZero")
        return x % y

    def pow(self, x, y): return pow(x, y, self.q)
    def band(self, x, y): return self._wrap(x & y)
    def bor(self, x, y): return self._wrap(x | y)
    def bxor(self, x, y): return self._wrap(x ^ y)
    def bnot(self, x): return self._wrap(~x & (self.R - 1))

    def shl(self, x, y):
        if y < self.bits:
            return self._wrap(x << y)
        k = self.q - y
        return 0 if k >= self.bits else x >> k

    def shr(self, x, y):
        if y < self.bits:
            return x >> y
        k = self.q - y
        return 0 if k >= self.bits else self._wrap(x << k)

    def eq(self, x, y): return int(x == y)
    def neq(self, x, y): return int(x != y)
    def lt(self, x, y): return int(self.val(x) < self.val(y))
    def gt(self, x, y): return int(self.val(x) > self.val(y))
    def leq(self, x, y): return int(self.val(x) <= self.val(y))
    def geq(self, x, y): return int(self.val(x) >= self.val(y))
    def land(self, x, y): return int(x != 0 and y != 0)
    def lor(self, x, y): return int(x != 0 or y != 0)
    def lnot(self, x): return int(x == 0)


_CACHE: dict = {}


def fp_for(prime: str) -> Fp:
    if prime not in _CACHE:
        _CACHE[prime] = Fp(PRIMES[prime], prime)
    return _CACHE[prime]
