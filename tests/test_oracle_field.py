"""Pins oracle/field.py (the Python restatement of circom's field semantics) against
 (1) the reference's own unit tests (toy prime 257, circom_algebra/src/modular_arithmetic.rs:217-269),
 (2) the reference's documented operator examples (mkdocs basic-operators.md),
 (3) the compiled reference field library (generic/fr.cpp rendered + g++) in every tagged
     representation, on random and edge operands, for bn128 and bls12381,
 (4) the Montgomery constants in the reference asm data sections."""
import random

import pytest

from oracle.field import Field, PRIMES, FieldError
from oracle.ref_shim import RefFr, BINOPS, REP_AUTO, REP_LONG, REP_MONT
from oracle import render_fr


def test_reference_unit_tests_prime_257():
    f = Field(257)
    # mod_check: modulus(-8, 5) == 2  (python % is already floor-mod)
    assert (-8) % 5 == 2
    # comparison_check: (2-1) != -1
    assert f.neq(f.sub(2, 1), f.norm(-1)) == 1
    # mod_operation_check: 17 % 32 == 17
    assert f.mod(17, 32) == 17
    # complement_of_complement_is_the_original_test
    assert f.bnot(f.bnot(1234 % 257)) == 1234 % 257
    # lesser_eq_test: 0 <= 2
    assert f.leq(0, 2) == 1


def test_documented_operator_examples():
    f = Field(PRIMES["bn128"])
    p = f.q
    # basic-operators.md:42-58: val(p-1) = -1 < val(1) = 1
    assert f.lt(p - 1, 1) == 1 and f.gt(1, p - 1) == 1
    # p/2+1 is negative, p/2 is positive (integer division)
    assert f.lt(p // 2 + 1, 0) == 1 and f.gt(p // 2, 0) == 1
    # shifts: basic-operators.md:103-117: x >> k = x/(2**k) for 0<=k<=p/2 ; x >> k = x << (p-k) otherwise
    assert f.shr(1 << 200, 100) == 1 << 100
    assert f.shr(5, p - 3) == f.shl(5, 3) == 40
    assert f.shl(5, p - 1) == 2
    # Multiplier2 example 3*11 = 33 (computing-the-witness.md)
    assert f.mul(3, 11) == 33


def test_montgomery_constants_match_asm_data_sections():
    # SURVEY Appendix C columns (= labels q, half, R2, R3, lboMask, np at the end of <prime>/fr.asm)
    want = {
        "bn128": dict(fr_np="0xc2e1f593efffffff", lboMask="0x3fffffffffffffff",
                      fr_r2_list=["0x1bb8e645ae216da7", "0x53fe3ab1e35c59e3", "0x8c49833d53bb8085", "0x216d0b17f4e44a5"],
                      half_list=["0xa1f0fac9f8000000", "0x9419f4243cdcb848", "0xdc2822db40c0ac2e", "0x183227397098d014"]),
        "bls12381": dict(fr_np="0xfffffffeffffffff", lboMask="0x7fffffffffffffff",
                         fr_r2_list=["0xc999e990f3f29c6d", "0x2b6cedcb87925c23", "0x5d314967254398f", "0x748d9d99f59ff11"],
                         half_list=["0x7fffffff80000000", "0xa9ded2017fff2dff", "0x199cec0404d0ec02", "0x39f6d3a994cebea4"]),
    }
    for prime, w in want.items():
        got = render_fr.params_for(PRIMES[prime])
        for k, v in w.items():
            assert got[k] == v, (prime, k)
        assert got["cannotOptimize"] is False
    assert render_fr.params_for(PRIMES["secq256r1"])["cannotOptimize"] is True


def _edge_values(f: Field):
    q = f.q
    e = [0, 1, 2, 3, 31, 32, 63, 64, 253, 254, 255, 256, f.half, f.half + 1, f.half - 1, q - 1, q - 2,
         (1 << 31) - 1, 1 << 31, (1 << 31) + 1, (1 << 32) - 1, 1 << 32, (1 << 63), (1 << 64) - 1, 1 << 64,
         q - (1 << 31), q - (1 << 31) - 1, q - (1 << 31) + 1, q - 253, q - 254, q - 255, q - 64,
         (1 << f.bits - 1), (1 << f.bits - 1) - 1, f.mask % q]
    return [x % q for x in e]


@pytest.mark.parametrize("prime", ["bn128", "bls12381", "pallas", "secq256r1"])
def test_field_py_matches_compiled_reference(prime, request):
    # bn128 / bls12381: the BASELINE primes; pallas: another 255-bit prime; secq256r1: the one prime whose top limb
    # forces the reference's `cannotOptimize` Montgomery variant (generic/fr.cpp:115-163)
    from conftest import ensure_ref
    ref_dir = ensure_ref(prime)
    ref = RefFr(ref_dir / "libfr_shim.so")
    f = Field(PRIMES[prime])
    assert ref.q == f.q
    rng = random.Random(1234)
    edges = _edge_values(f)
    rand = [rng.randrange(f.q) for _ in range(40)] + [rng.randrange(1 << 31) for _ in range(10)] + \
           [f.q - rng.randrange(1, 1 << 31) for _ in range(10)] + [rng.randrange(1 << 70) for _ in range(10)]
    pairs = [(a, b) for a in edges for b in edges] + [(rng.choice(rand), rng.choice(rand + edges)) for _ in range(1500)]
    n = 0
    for a, b in pairs:
        for name in BINOPS:
            if name in ("idiv", "mod") and b == 0:
                with pytest.raises(FieldError):
                    getattr(f, name)(a, b)
                continue
            if name == "pow" and b.bit_length() > 64 and n % 50:
                continue  # long exponents are slow in the reference; sample them
            want = getattr(f, name)(a, b)
            for ra, rb in ((REP_AUTO, REP_AUTO), (REP_LONG, REP_MONT), (REP_MONT, REP_MONT), (REP_MONT, REP_AUTO)):
                got = ref.binop(name, a, b, ra, rb)
                assert got == want, (prime, name, hex(a), hex(b), ra, rb, hex(got), hex(want))
            n += 1
    for a in edges + rand:
        for name in ("neg", "bnot", "lnot", "inv"):
            for ra in (REP_AUTO, REP_LONG, REP_MONT):
                assert ref.unop(name, a, ra) == getattr(f, name)(a), (prime, name, hex(a), ra)
        assert ref.unop("square", a, REP_MONT) == f.mul(a, a)
        assert ref.is_true(a) == f.is_true(a)
        try:
            want = f.to_int(a)
        except FieldError:
            continue
        assert ref.to_int(a) == want
    # raw Montgomery product and string ingest
    for _ in range(200):
        a, b = rng.randrange(f.q), rng.randrange(f.q)
        assert ref.raw_mmul(a, b) == f.mmul(a, b)
    for s, base in (("123456789012345678901234567890123456789012345678901234567890123456789012345678901234567890", 10),
                    ("ff" * 40, 16), ("1011" * 70, 2), ("7654321" * 20, 8), ("0", 10), ("-5", 10)):
        assert ref.str2element(s, base) == f.from_str(s, base)


def test_device_inversion_model_matches_pow():
    """oracle/bingcd_model.py restates the device's constant-time binary-GCD inverse limb for limb (with the
    identities it relies on as assertions); it must agree with pow(y, -1, q) and map 0 to 0 like mpz_invert."""
    import random
    from oracle.bingcd_model import inv_mod
    rng = random.Random(3)
    for name, q in PRIMES.items():
        if q.bit_length() < 200:
            continue
        vals = [0, 1, 2, 3, q - 1, q - 2, (q - 1) // 2, (q + 1) // 2, 1 << 64, (1 << 64) - 1, 1 << 128, (1 << 253) % q]
        vals += [rng.randrange(q) for _ in range(150)] + [rng.randrange(1 << 40) for _ in range(30)]
        for y in vals:
            assert inv_mod(y, q) == (pow(y, -1, q) if y else 0), (name, hex(y))
