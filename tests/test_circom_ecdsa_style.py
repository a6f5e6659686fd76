"""Big-integer arithmetic written in the STYLE of circom-ecdsa (circuits/circomlib/bigint_ecdsa.circom: unit adders with
explicit carries, carry-less products pinned by a polynomial identity, registers split by hints, a long-division FUNCTION over
100- and 200-entry arrays whatever k is) - the idioms of the library BASELINE config 5 is built from: BigMultModP against
plain integers, every constraint satisfied, and what the over-allocated arrays cost once the function is bytecode."""
import random

from circom_amd.frontend.circom_exec import build_program
from circom_amd.frontend.circom_lang import parse_program
from circom_amd.frontend.flatten import flatten
from oracle.tape_eval import check_r1cs, eval_flat
from tests.test_circom_frontend import LIB


def test_bigmultmodp_in_library_style(tmp_path):
    n, k = 28, 3
    f = tmp_path / "bms.circom"
    f.write_text('include "bigint_ecdsa.circom";\ncomponent main = BigMultModPStyle(%d, %d);\n' % (n, k))
    fc = flatten(build_program(parse_program(str(f), [LIB]), "bls12381"))
    assert (fc.n_signals, len(fc.constraints)) == (1178, 1191)
    fn, = fc.functions
    # the same division with exact array sizes (bigint_func.circom long_div) is 1 229 instructions on 45 registers.  Only the
    # elements a run-time region assigns live in registers and a function with one exit returns its value without result
    # registers - with whole arrays pinned and copied out this function was 9 903 instructions on 1 230 registers
    assert fn["name"] == "e_long_div$0" and len(fn["code"]) < 6000 and fn["n_regs"] < 400
    rng = random.Random(2)
    lim = lambda x: [(x >> (n * i)) & ((1 << n) - 1) for i in range(k)]
    for t in range(8):
        p = (rng.getrandbits(n * k) | (1 << (n * k - 1))) if t % 2 else (rng.getrandbits(n * k - 20) | (1 << (n * (k - 1))))
        a, b = (rng.randrange(p), rng.randrange(p)) if t < 6 else (p - 1, p - 1)
        inp = {fc.main_input_start + i: v for i, v in enumerate(lim(a) + lim(b) + lim(p))}
        sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp, functions=fc.functions)
        assert failed is None
        assert sum(v << (n * i) for i, v in enumerate(sig[1:1 + k])) == a * b % p
        assert check_r1cs(fc.fp.q, fc.constraints, sig) is None
