"""BASELINE config 5: secp256k1 ECDSA verification as a circuit over the BLS12-381 scalar field (circuits/secp256k1.py over
circuits/bigint_func.py, in the shape of 0xPARC circom-ecdsa).

CPU: the big-integer FUNCTIONS as tier-2 bytecode equal their native closed forms (the oracle's interpreter executes the real
body: Fermat inverse as a run-time loop of 256 trips); every template on a toy curve (16-bit prime, prime order) against host
curve arithmetic, with every constraint satisfied; the reference C++ RUNTIME executes the emitted C++ of the toy verifier -
function bodies included - and writes the oracle's `.wtns` byte for byte; the full-size verifier (2.47 M signals, 2.49 M
constraints) accepts a valid signature and rejects a corrupted one.  GPU: the toy verifier with INTERPRETED functions, and the
full-size curve operations with the NATIVE device routines (binary-GCD inverse + Montgomery products modulo the secp256k1
prime), both against the oracle; the full verifier at its batch lives in tests/test_baseline_configs.py."""
import random

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.field import PRIMES
from circom_amd.frontend.dsl import Program
from circom_amd.frontend.flatten import flatten
from circom_amd.frontend.rtcode import RtFunction
from circom_amd.circuits import secp256k1 as S, bigint_func as BF
from circom_amd.hip_elements.writers import wtns_bytes
from oracle.field import Field
from oracle import tape_eval as TE
from oracle.tape_eval import eval_flat, check_r1cs

TOY = (65419, 64921, 3, 48428)          # y^2 = x^3 + 7 over a 16-bit prime, prime group order, generator (3, 48428)
TN, TK = 8, 2
Q = PRIMES["bls12381"]


class _FP:
    q = Q


def _run_bytecode(fn, args):
    """execute the function's real body in the oracle's interpreter (no closed form)"""
    consts, code = [], []
    for ins in fn.code:
        ins = list(ins)
        for j in (2, 3):
            x = ins[j]
            if isinstance(x, tuple) and len(x) == 2 and x[0] == 'c':
                consts.append(x[1])
                ins[j] = ('c', len(consts) - 1)
        code.append(tuple(ins))
    d = fn.as_data()
    d["code"], d["native"] = code, None
    regs = list(args) + [0] * (fn.n_regs - len(args))
    assert TE.run_function(Field(Q), d, regs, 0, consts)
    return regs[fn.ret_base:fn.ret_base + fn.n_ret]


def _point(cv, rnd):
    return S.ec_mul(cv, rnd.randrange(1, cv[1]), (cv[2], cv[3]))


def _limbs(P, n, k):
    return BF.limbs_of(P[0], n, k) + BF.limbs_of(P[1], n, k)


@pytest.mark.parametrize("cv,n,k", [(TOY, TN, TK), (S.SECP256K1, 64, 4)])
def test_function_bodies_equal_their_closed_forms(cv, n, k):
    """mod_inv / ec_add / ec_double: bytecode (schoolbook prod, Knuth long_div, square-and-multiply with a run-time indexed
    exponent limb) == Python integers, on random arguments and on the edges 0, 1, p - 1"""
    p = cv[0]
    rnd = random.Random(n)
    f_inv = RtFunction("mod_inv", k, BF.build_mod_inv(n, k, p), _FP)
    f_add = RtFunction("ec_add", 4 * k, BF.build_ec_add_unequal(n, k, p), _FP)
    f_dbl = RtFunction("ec_double", 2 * k, BF.build_ec_double(n, k, p), _FP)
    for a in [0, 1, p - 1, rnd.randrange(p)] + ([rnd.randrange(p) for _ in range(4)] if n < 64 else []):
        args = BF.limbs_of(a, n, k)
        assert _run_bytecode(f_inv, args) == BF.native_eval("mod_inv", n, k, p, args)
        assert BF.int_of(BF.native_eval("mod_inv", n, k, p, args), n) * a % p == (1 if a else 0)
    for _ in range(3 if n < 64 else 1):
        P1, P2 = _point(cv, rnd), _point(cv, rnd)
        args = _limbs(P1, n, k) + _limbs(P2, n, k)
        got = _run_bytecode(f_add, args)
        assert got == BF.native_eval("ec_add", n, k, p, args) and got[k:] == _limbs(S.ec_add(cv, P1, P2), n, k)
        got = _run_bytecode(f_dbl, _limbs(P1, n, k))
        assert got == BF.native_eval("ec_double", n, k, p, _limbs(P1, n, k)) and got[k:] == _limbs(S.ec_add(cv, P1, P1), n, k)


def _eval(fc, inputs):
    sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code,
                            {fc.main_input_start + i: v for i, v in enumerate(inputs)}, fc.functions)
    return sig, failed


def test_toy_curve_templates_against_host_arithmetic():
    rnd = random.Random(2)
    n, k, cv = TN, TK, TOY
    G = (cv[2], cv[3])
    P1, P2 = _point(cv, rnd), _point(cv, rnd)
    for prog, inputs, want in (
            (Program(S.EcAddUnequal(n, k, cv), prime="bls12381"), _limbs(P1, n, k) + _limbs(P2, n, k), S.ec_add(cv, P1, P2)),
            (Program(S.EcDouble(n, k, cv), prime="bls12381"), _limbs(P1, n, k), S.ec_add(cv, P1, P1))):
        fc = flatten(prog)
        sig, failed = _eval(fc, inputs)
        assert failed is None and sig[1:1 + 2 * k] == _limbs(want, n, k)
        assert check_r1cs(fc.fp.q, fc.constraints, sig) is None
    fc = flatten(Program(S.EcScalarMult(n, k, cv), prime="bls12381"))
    for s in (1, 2, 3, 0x8001, 40000, rnd.randrange(1, cv[1] - 1)):
        sig, failed = _eval(fc, BF.limbs_of(s, n, k) + _limbs(P1, n, k))
        assert failed is None and sig[1:1 + 2 * k] == _limbs(S.ec_mul(cv, s, P1), n, k), s
        assert check_r1cs(fc.fp.q, fc.constraints, sig) is None
    fc = flatten(Program(S.EcFixedBaseMult(n, k, cv, 4), prime="bls12381"))
    for s in (1, 5, 16, 0x0F0, 0x1000, 0x1001, 54321):          # zero digits in every position
        sig, failed = _eval(fc, BF.limbs_of(s, n, k))
        assert failed is None and sig[1:1 + 2 * k] == _limbs(S.ec_mul(cv, s, G), n, k), s
        assert check_r1cs(fc.fp.q, fc.constraints, sig) is None


def test_toy_verifier_accepts_valid_and_rejects_corrupted_signatures():
    rnd = random.Random(3)
    fc = flatten(Program(S.ECDSAVerifyNoPubkeyCheck(TN, TK, TOY, 4), prime="bls12381"))
    for _ in range(4):
        inp = S.sign(TOY, TN, TK, rnd)
        sig, failed = _eval(fc, inp)
        assert failed is None and sig[1] == 1 and check_r1cs(fc.fp.q, fc.constraints, sig) is None
        bad = list(inp)
        bad[2 * TK] ^= 1                                          # another message hash: still a satisfiable witness, result 0
        sig, failed = _eval(fc, bad)
        assert failed is None and sig[1] == 0 and check_r1cs(fc.fp.q, fc.constraints, sig) is None


def test_reference_runtime_executes_the_toy_verifier(tmp_path):
    """the reference C++ runtime runs the emitted C++ of the toy verifier (function BODIES as labels + gotos over its own Fr_*
    calls: the native closed forms are nowhere in that binary) and writes the `.wtns` the oracle predicts WITH its closed forms"""
    ref_build = pytest.importorskip("oracle.ref_build")
    if not ref_build.REF_ROOT.exists():
        pytest.skip("reference tree not present")
    cp = compile_program(Program(S.ECDSAVerifyNoPubkeyCheck(TN, TK, TOY, 4), prime="bls12381"), str(tmp_path), "ecdsa_toy", sym=False, strands=(1,))
    ref_build.build_circuit(cp)
    rnd = random.Random(4)
    rows = [S.sign(TOY, TN, TK, rnd) for _ in range(3)]
    raw = b"".join(int(v).to_bytes(32, "little") for r in rows for v in r)
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=str(tmp_path / "r_"))
    for i, r in enumerate(rows):
        sig, failed = _eval(cp.flat, r)
        assert failed is None and sig[1] == 1
        assert (tmp_path / ("r_%d.wtns" % i)).read_bytes() == wtns_bytes(cp.flat.fp.q, sig)


def test_full_size_verifier_on_the_oracle():
    """n = 64, k = 4, secp256k1, stride 8: the size BASELINE config 5 names (its "~1.5 M constraints": 2.49 M here - the slope
    is a range-checked signal and every modular relation pays its own quotient and carries)"""
    import hashlib
    import json
    import os
    rnd = random.Random(5)
    fc = flatten(Program(S.ECDSAVerifyNoPubkeyCheck(64, 4, S.SECP256K1, 8), prime="bls12381"))
    assert fc.n_signals > 2_000_000 and len(fc.constraints) > 2_000_000
    inp = S.sign(S.SECP256K1, 64, 4, rnd)
    sig, failed = _eval(fc, inp)
    assert failed is None and sig[1] == 1
    assert check_r1cs(fc.fp.q, fc.constraints, sig) is None
    # golden vectors written by the REFERENCE RUNTIME (tests/golden/make_golden.py ecdsa: it executed the bodies of the witness
    # functions); the oracle - with the closed forms - must reproduce the 79 MB files: a valid signature and the rejected one
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_wtns_ecdsa.json")))
    assert gold["n_signals"] == fc.n_signals and gold["n_constraints"] == len(fc.constraints)
    for vec in (gold["cases"]["ecdsa_verify"]["vectors"][1], gold["cases"]["ecdsa_verify"]["vectors"][3]):
        sig, failed = _eval(fc, [int(v) for v in vec["inputs"]])
        assert failed is None and [str(v) for v in sig[:8]] == vec["witness_head"]
        b = wtns_bytes(fc.fp.q, sig)
        assert len(b) == vec["wtns_len"] and hashlib.sha256(b).hexdigest() == vec["wtns_sha256"]
    assert gold["cases"]["ecdsa_verify"]["vectors"][3]["witness_head"][1] == "0"


@pytest.mark.gpu
def test_gpu_toy_verifier_with_interpreted_functions(tmp_path, monkeypatch):
    """a 16-bit prime is outside the device's field code: the function bodies run in the per-lane interpreter (run-time loops,
    run-time indexed limbs, divergent lanes) - on one strand (what cw_batch_create picks: the calls are one serial chain, more
    strands would only wait) and, forced, on 16 strands (calls as heavy units between FULL barriers, D_BITS rows)"""
    from circom_amd import runtime as rt
    cp = compile_program(Program(S.ECDSAVerifyNoPubkeyCheck(TN, TK, TOY, 4), prime="bls12381"), str(tmp_path), "ecdsa_toy", sym=False, strands=(1, 16))
    assert all(f[2] is None for f in cp.tape.functions)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    rnd = random.Random(6)
    B = 96
    rows = [S.sign(TOY, TN, TK, rnd) for _ in range(B)]
    for i in range(0, B, 7):
        rows[i][2 * TK] ^= 1                                      # invalid signatures among them
    seen = []
    for want in (None, 16, 1):
        if want is None:
            monkeypatch.delenv("CW_STRANDS", raising=False)
        else:
            monkeypatch.setenv("CW_STRANDS", str(want))
        b = c.batch(B)
        assert want is None or b.strands == want
        seen.append(b.strands)
        b.set_inputs(rows)
        b.run(); b.check_r1cs(); b.sync()
        assert (b.status() == 0).all()
        for i in (0, 1, 7, 50, 95):
            sig, failed = _eval(cp.flat, rows[i])
            assert failed is None and b.witness(i) == sig and sig[1] == (0 if i % 7 == 0 else 1)
        b.close()
    assert set(seen) == {1, 16}
    c.close()


@pytest.mark.gpu
def test_gpu_native_curve_functions_on_secp256k1(tmp_path):
    """EcAddUnequal / EcDouble / BigModInvConst at full size: the hints come from the device's native routines (the tape says
    so), every witness equals the oracle's, every constraint holds"""
    from circom_amd import runtime as rt
    cv, n, k = S.SECP256K1, 64, 4
    rnd = random.Random(7)
    B = 130
    for name, prog, mk in (
            ("ecadd", Program(S.EcAddUnequal(n, k, cv), prime="bls12381"), lambda: _limbs(_point(cv, rnd), n, k) + _limbs(_point(cv, rnd), n, k)),
            ("ecdbl", Program(S.EcDouble(n, k, cv), prime="bls12381"), lambda: _limbs(_point(cv, rnd), n, k)),
            ("modinv", Program(S.BigModInvConst(n, k, cv[1]), prime="bls12381"), lambda: BF.limbs_of(rnd.randrange(cv[1]), n, k))):
        cp = compile_program(prog, str(tmp_path), name, sym=False, strands=(1,))
        assert any(f[2] is not None for f in cp.tape.functions)
        c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
        rows = [mk() for _ in range(B)]
        if name == "modinv":
            rows[3] = BF.limbs_of(0, n, k)                       # inverse of 0 = 0: `in * out = 1` then fails in the CHECK, not in the hint
        b = c.batch(B)
        b.set_inputs(rows)
        b.run(); b.check_r1cs(); b.sync()
        st = b.status()
        for i in range(B):
            sig, failed = _eval(cp.flat, rows[i])
            assert b.witness(i) == sig, (name, i)
            assert (st[i] & 1) == (0 if failed is None else 1), (name, i)
            assert bool(st[i] & 4) == (check_r1cs(cp.flat.fp.q, cp.flat.constraints, sig) is not None), (name, i)
        b.close(); c.close()


# ---- long_div with its closed form (the quotient hint of CheckZeroModP) ----------------------------------------------------------
def _long_div_cases(n, k, m, rnd, count):
    """argument vectors inside the tag's contract (proper limbs, b[k-1] != 0), with the edges Knuth D cares about: quotient
    digits of all ones, a divisor of minimal / maximal top limb, a dividend just below and at a multiple of the divisor"""
    top = (1 << n) - 1
    out = []
    for it in range(count):
        kind = it % 6
        if kind == 0:
            b = rnd.randrange(1 << (n * (k - 1)), 1 << (n * k))
        elif kind == 1:
            b = (1 << (n * (k - 1))) + rnd.randrange(1 << (n * (k - 1)))            # smallest top limb
        elif kind == 2:
            b = (1 << (n * k)) - 1 - rnd.randrange(1 << 8)                          # all-ones divisor
        else:
            b = rnd.randrange(1 << (n * k - 1), 1 << (n * k))
        if kind == 3:
            a = b * rnd.randrange(1 << (n * m)) - 1                                 # remainder b - 1
        elif kind == 4:
            a = b * rnd.randrange(1 << (n * m))                                     # remainder 0
        elif kind == 5:
            a = (1 << (n * (k + m))) - 1 - rnd.randrange(1 << 16)                   # the largest dividends
        else:
            a = rnd.randrange(1 << (n * (k + m)))
        a = max(a, 0) % (1 << (n * (k + m)))
        out.append(BF.limbs_of(a, n, k + m) + BF.limbs_of(b, n, k))
    out.append([top] * (k + m) + [0] * (k - 1) + [1])                               # b = 2^(n (k-1)): pure limb shift
    out.append([0] * (k + m) + [top] * k)
    return out


@pytest.mark.parametrize("n,k,m", [(8, 2, 3), (64, 4, 5), (32, 3, 3), (50, 3, 2)])
def test_long_div_body_equals_its_closed_form(n, k, m):
    rnd = random.Random(n * 100 + k * 10 + m)
    f_div = RtFunction("long_div", 2 * k + m, BF.build_long_div(n, k, m), _FP)
    assert f_div.ret_base == 2 * k + m and f_div.n_ret == m + 1 + k
    for args in _long_div_cases(n, k, m, rnd, 6 if n >= 50 else 18):
        got = _run_bytecode(f_div, args)
        assert got == BF.native_eval("long_div", n, k, m, args)
        a, b = BF.int_of(args[:k + m], n), BF.int_of(args[k + m:], n)
        assert BF.int_of(got[:m + 1], n) * b + BF.int_of(got[m + 1:], n) == a and BF.int_of(got[m + 1:], n) < b
    with pytest.raises(ZeroDivisionError):                       # outside the contract the closed form refuses (the oracle then runs the body)
        BF.native_eval("long_div", n, k, m, [1] * (k + m) + [1] * (k - 1) + [0])


def _long_div_program(n, k, m, native=True):
    from circom_amd.frontend.dsl import template

    @template
    def LongDivHint(c, n, k, m):
        a = c.input("a", k + m)
        b = c.input("b", k)
        out = c.output("out", m + 1 + k)
        fn = c.function("long_div_%d_%d_%d" % (n, k, m), 2 * k + m, BF.build_long_div(n, k, m),
                        native=("long_div", n, k, m) if native else None)
        res = c.call(fn, [a[i] for i in range(k + m)] + [b[i] for i in range(k)])
        for i in range(m + 1 + k):
            c.hint(out[i], res[i])
    return Program(LongDivHint(n, k, m), prime="bls12381")


def test_long_div_tag_reaches_the_tape(tmp_path):
    cp = compile_program(_long_div_program(64, 4, 5), str(tmp_path), "ld", sym=False, strands=(1,), fpjit=False)
    assert [f[2] for f in cp.tape.functions] == [(4, 64, 4, 5)]
    from circom_amd import runtime as rt
    rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path).close()     # the loader accepts the tag (shape checks of cw_load)
    cp = compile_program(_long_div_program(8, 2, 3), str(tmp_path), "ld8", sym=False, strands=(1,), fpjit=False)
    assert [f[2] for f in cp.tape.functions] == [None]           # limbs below 32 bits: the device interprets the body
    # the schedule replay (oracle of the device's rows) takes the closed form as well
    from oracle.tape_eval import eval_tape
    cp = compile_program(_long_div_program(32, 3, 3), str(tmp_path), "ld32", sym=False, strands=(1,), fpjit=False)
    rnd = random.Random(5)
    for args in _long_div_cases(32, 3, 3, rnd, 6):
        sig, st = eval_tape(cp.tape, {cp.flat.main_input_start + i: v for i, v in enumerate(args)})
        assert st == 0 and sig[1:1 + 7] == BF.native_eval("long_div", 32, 3, 3, args)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,m", [(64, 4, 5), (32, 3, 3), (50, 3, 2), (64, 4, 4), (64, 2, 7)])
def test_gpu_native_long_div(tmp_path, n, k, m):
    """the device's Knuth D on packed 32-bit words (csrc/cw_call.hip.h eval_call_long_div: word-aligned and generic limb
    widths) against Python integers, one argument vector per lane incl. the edge cases; one lane outside the contract
    (top limb of the divisor 0) is flagged instead of answered"""
    from circom_amd import runtime as rt
    cp = compile_program(_long_div_program(n, k, m), str(tmp_path), "ld", sym=False, strands=(1,))
    assert cp.tape.functions[0][2] == (4, n, k, m)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    rnd = random.Random(n + k + m)
    rows = _long_div_cases(n, k, m, rnd, 190)
    bad = 17
    rows[bad] = rows[bad][:k + m] + rows[bad][k + m:2 * k + m - 1] + [0]
    b = c.batch(len(rows))
    b.set_inputs(rows)
    b.run(); b.sync()
    st = b.status()
    for i, args in enumerate(rows):
        if i == bad:
            assert st[i] & 3
            continue
        assert st[i] == 0, i
        assert b.witness(i)[1:1 + m + 1 + k] == BF.native_eval("long_div", n, k, m, args), (i, args)
    b.close(); c.close()


@pytest.mark.parametrize("which", ["add", "double", "verify"])
def test_schedules_with_calls_on_several_strands(which):
    """circuits with run-time functions on 1 / 4 / 16 strands: a call is a heavy unit of its level between FULL barriers, the
    bit rows of the range checks travel as D_BITS rows (lower.py passes A7 / C) - the replay of the rows (oracle/tape_eval.py,
    which reports every cross-strand hand-off without the barrier it needs and fetches operands one row ahead like the kernel)
    gives the flat program's witness"""
    from circom_amd.hip_elements.lower import lower, D_BITS
    from oracle.tape_eval import eval_tape
    rnd = random.Random(11)
    n, k, cv = TN, TK, TOY
    P1, P2 = _point(cv, rnd), _point(cv, rnd)
    if which == "add":
        prog, inputs = Program(S.EcAddUnequal(n, k, cv), prime="bls12381"), _limbs(P1, n, k) + _limbs(P2, n, k)
    elif which == "double":
        prog, inputs = Program(S.EcDouble(n, k, cv), prime="bls12381"), _limbs(P1, n, k)
    else:
        prog, inputs = Program(S.ECDSAVerifyNoPubkeyCheck(n, k, cv, 2), prime="bls12381"), S.sign(cv, n, k, rnd)
    fc = flatten(prog)
    inp = {fc.main_input_start + i: v for i, v in enumerate(inputs)}
    want, failed = _eval(fc, inputs)
    assert failed is None
    for S_ in (1, 4, 16):
        t = lower(fc, n_strands=S_)
        t1 = t
        if S_ == 1:
            assert t.stats["bits_rows"] == 0                  # (one strand keeps the rows the emitted code has bodies for)
            t = lower(fc, n_strands=1, fuse_bits=True)
        assert t.n_strands == S_ and t.stats["bits_rows"] > 0 and t.stats["bits_fused"] > 2 * t.stats["bits_rows"]
        for tt in {id(t): t, id(t1): t1}.values():
            got, st = eval_tape(tt, inp)
            assert st == 0 and got == want, S_
        if S_ > 1:
            assert t.stats["full_barriers"] > 0
    # without the fusion: the same witness from one row per bit
    t = lower(fc, n_strands=4, fuse_bits=False)
    assert t.stats["bits_rows"] == 0
    got, st = eval_tape(t, inp)
    assert st == 0 and got == want
