"""Functions compiled from circom text to tier-2 bytecode, run with POISONED registers: on the device a function's registers
are temporaries of the value table that hold whatever the previous rows left there, the oracle's evaluator zero-fills them -
a read of a register that no instruction wrote on that path would go unnoticed here and differ there.  Every register except the
arguments starts as None: any arithmetic on it raises."""
import os
import random

import pytest

from circom_amd.frontend.circom_exec import build_program, program_from_text
from circom_amd.frontend.circom_lang import parse_program
from circom_amd.frontend.flatten import flatten
from oracle.field import Field, PRIMES
from oracle.tape_eval import run_function
from tests.test_circom_frontend import LIB, RT_SRC

Q = PRIMES["bls12381"]


def _call(fc, fn, args):
    regs = [None] * fn["n_regs"]
    regs[:len(args)] = [a % fc.fp.q for a in args]
    assert run_function(Field(fc.fp.q), fn, regs, 0, fc.constants)
    out = regs[fn["ret_base"]:fn["ret_base"] + fn["n_ret"]]
    assert all(isinstance(v, int) for v in out), out
    return out


def _functions(tmp_path, text, prime="bls12381"):
    f = tmp_path / "p.circom"
    f.write_text(text)
    fc = flatten(build_program(parse_program(str(f), [LIB]), prime))
    return fc, {fn["name"].split("$")[0]: fn for fn in fc.functions}


def test_long_division_and_fermat_inverse_never_read_an_unwritten_register(tmp_path):
    rng = random.Random(9)
    n, k = 28, 3
    lim = lambda x, kk=k: [(x >> (n * i)) & ((1 << n) - 1) for i in range(kk)]
    val = lambda l: sum(v << (n * i) for i, v in enumerate(l))
    fc, fns = _functions(tmp_path, 'include "bigint_ecdsa.circom";\ncomponent main = BigMultModPStyle(%d, %d);\n' % (n, k))
    for t in range(6):
        p = (rng.getrandbits(n * k) | (1 << (n * k - 1))) if t % 2 else (rng.getrandbits(n * k - 30) | (1 << (n * (k - 1))))
        a = rng.getrandbits(2 * n * k - 3) % (p << (n * k - 2)) if t else p * p - 1
        a %= p * (1 << (n * k))                               # the quotient fits k + 1 registers
        out = _call(fc, fns["e_long_div"], lim(a, 2 * k) + lim(p))
        # out[2][100] flattened: quotient registers first, then the remainder's
        assert val(out[:k + 1]) == a // p and val(out[100:100 + k]) == a % p
    n, k = 16, 2
    fc, fns = _functions(tmp_path, open(os.path.join(os.path.dirname(__file__), "circom", "modinv.circom")).read())
    p = 2147483647
    for a in (1, 2, p - 1, 123456789, 0):
        out = _call(fc, fns["mod_inv"], lim(a, 2) + lim(p, 2))
        assert val(out) == (pow(a, p - 2, p) if a else 0)


def test_loops_indices_and_early_returns_with_poisoned_registers():
    fc = flatten(program_from_text(RT_SRC))
    fns = {fn["name"].split("$")[0]: fn for fn in fc.functions}
    for x in (0, 1, 26, 99, 1000):
        r = 0
        while (r + 1) * (r + 1) <= x:
            r += 1
        assert _call(fc, fns["isqrt"], [x]) == [r]
    assert _call(fc, fns["collatz"], [27]) == [111]
    for i in range(6):
        assert _call(fc, fns["pick"], [5, i]) == [[5, 25, 7, 105][i] if i < 4 else 0]
    assert _call(fc, fns["hist"], [3, 4]) == [1, 1, 1]        # hist(x, x + i, 5): the third argument is a constant of the call site
