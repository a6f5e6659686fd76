"""CPU oracle: prime-field operator semantics of circom, restated on Python ints.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import anything under oracle/.

Every operator is defined on canonical values in [0, q).  Sources followed:
  * run-time truth : code_producers/src/c_elements/generic/fr.cpp  (Fr_* functions)
  * compile-time twin: circom_algebra/src/modular_arithmetic.rs
(SURVEY.md Appendix D tabulates both.)  Where they differ (division by zero) the
*run-time* behaviour is what the witness calculator exhibits and what is modelled.

Pinned by tests/test_oracle_field.py against
  - the toy-prime-257 unit tests of modular_arithmetic.rs:217-269,
  - the compiled reference library (oracle/_ref/<prime>/libfr_shim.so) on random and
    edge operands in every tagged representation.
"""
from __future__ import annotations

# program_structure/src/utils/constants.rs:3-13
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


class FieldError(Exception):
    """Raised where the reference run time aborts (GMP division by zero, Fr_toInt overflow)."""


class Field:
    def __init__(self, q: int):
        self.q = q
        self.bits = q.bit_length()                 # {{qbits}} in generic/fr.cpp
        self.mask = (1 << self.bits) - 1           # lboMask applied to the top limb (fr.cpp:293-327)
        self.half = q >> 1                         # `half` (fr.cpp:9); val(x) = x-q iff x > half
        self.n64 = (self.bits + 63) // 64
        self.R = 1 << (64 * self.n64)              # Montgomery radix (fr.cpp:110-164)

    # ---- helpers -------------------------------------------------------------------
    def norm(self, x: int) -> int:
        return x % self.q

    def val(self, x: int) -> int:
        """Signed view used by relational operators (modular_arithmetic.rs:154-161, fr.cpp rltL1L2 :1208)."""
        return x - self.q if x > self.half else x

    # ---- arithmetic ---------------------------------------------------------------
    def add(self, x, y):  # Fr_add fr.cpp:1017 ; modular_arithmetic.rs:26
        return (x + y) % self.q

    def sub(self, x, y):  # Fr_sub :827 ; :36
        return (x - y) % self.q

    def mul(self, x, y):  # Fr_mul :559 ; :31
        return (x * y) % self.q

    def neg(self, x):     # Fr_neg :1372 ; prefix_sub :66
        return (-x) % self.q

    def inv(self, x):     # Fr_inv :2895 — mpz_invert failure leaves 0
        if x % self.q == 0:
            return 0
        return pow(x, -1, self.q)

    def div(self, x, y):  # Fr_div :2908 = mul(x, inv(y)); y == 0 -> 0 at run time
        return (x * self.inv(y)) % self.q

    def idiv(self, x, y):  # Fr_idiv :2835 mpz_fdiv_q on canonical values
        if y == 0:
            raise FieldError("integer division by zero (reference: GMP abort)")
        return (x // y) % self.q

    def mod(self, x, y):   # Fr_mod :2859 mpz_fdiv_r
        if y == 0:
            raise FieldError("modulo by zero (reference: GMP abort)")
        return (x % y) % self.q

    def pow(self, x, y):   # Fr_pow :2877 mpz_powm, exponent = canonical y ; 0^0 = 1
        return pow(x, y, self.q)

    # ---- bitwise ------------------------------------------------------------------
    def _wrap(self, v):    # one conditional subtraction after masking (fr.cpp:297-301)
        v &= self.mask
        return v - self.q if v >= self.q else v

    def band(self, x, y):  # Fr_band :1938, raw :293
        return self._wrap(x & y)

    def bor(self, x, y):   # Fr_bor :2489
        return self._wrap(x | y)

    def bxor(self, x, y):  # Fr_bxor :2678
        return self._wrap(x ^ y)

    def bnot(self, x):     # Fr_bnot :2730, raw :366 ; complement modular_arithmetic.rs:94-109
        return self._wrap(~x & ((1 << (64 * self.n64)) - 1))

    def shl(self, x, y):   # Fr_shl :2265 ; shift_l :111-123
        if y < self.bits:
            return self._wrap(x << y)
        k = self.q - y     # "big shift": negative shift amount flips direction (:2233-2263)
        if k >= self.bits:
            return 0
        return x >> k

    def shr(self, x, y):   # Fr_shr :2189 ; shift_r :124-136
        if y < self.bits:
            return x >> y
        k = self.q - y
        if k >= self.bits:
            return 0
        return self._wrap(x << k)

    # ---- relational / boolean (results are the integers 0/1) -------------------------
    def eq(self, x, y):    # Fr_eq :1469
        return int(x == y)

    def neq(self, x, y):   # Fr_neq :1503
        return int(x != y)

    def lt(self, x, y):    # Fr_lt :1350 (rlt :1294)
        return int(self.val(x) < self.val(y))

    def gt(self, x, y):    # Fr_gt :1755
        return int(self.val(x) > self.val(y))

    def leq(self, x, y):   # Fr_leq :1761
        return int(self.val(x) <= self.val(y))

    def geq(self, x, y):   # Fr_geq :1356
        return int(self.val(x) >= self.val(y))

    def land(self, x, y):  # Fr_land :1771 — not short-circuit
        return int(x != 0 and y != 0)

    def lor(self, x, y):   # Fr_lor :1540
        return int(x != 0 or y != 0)

    def lnot(self, x):     # Fr_lnot :1568
        return int(x == 0)

    def is_true(self, x):  # Fr_isTrue :1086
        return x != 0

    def to_int(self, x):   # Fr_toInt :1146 with Fr_longNormal/longNeg :1102-1143
        if x < (1 << 31):
            return x
        if self.q - x <= (1 << 31):
            return x - self.q
        raise FieldError("Fr_toInt: value does not fit an int (reference: assert(false))")

    # ---- string ingest (Fr_str2element :2805): int(s, base) floor-mod q -------------
    def from_str(self, s: str, base: int = 10) -> int:
        return int(s, base) % self.q

    # ---- Montgomery helpers (fr.cpp:110-255) -----------------------------------------
    def to_mont(self, x):
        return (x * self.R) % self.q

    def from_mont(self, x):
        return (x * pow(self.R, -1, self.q)) % self.q

    def mmul(self, a, b):
        """Raw Montgomery product a*b*R^-1 mod q (Fr_rawMMul)."""
        return (a * b * pow(self.R, -1, self.q)) % self.q

    # dispatch table by operator name (names follow compute_bucket.rs:7-34 / Fr_* symbols)
    BINOPS = ("add", "sub", "mul", "div", "idiv", "mod", "pow", "shl", "shr", "band", "bor",
              "bxor", "eq", "neq", "lt", "gt", "leq", "geq", "land", "lor")
    UNOPS = ("neg", "bnot", "lnot", "inv")


BN128 = Field(PRIMES["bn128"])
BLS12381 = Field(PRIMES["bls12381"])


def field_for(name: str) -> Field:
    return Field(PRIMES[name])
