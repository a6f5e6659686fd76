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
def test_gpu_toy_verifier_with_interpreted_functions(tmp_path):
    """a 16-bit prime is outside the device's field code: the function bodies run in the per-lane interpreter (run-time loops,
    run-time indexed limbs, divergent lanes)"""
    from circom_amd import runtime as rt
    cp = compile_program(Program(S.ECDSAVerifyNoPubkeyCheck(TN, TK, TOY, 4), prime="bls12381"), str(tmp_path), "ecdsa_toy", sym=False, strands=(1,))
    assert all(f[2] is None for f in cp.tape.functions)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    rnd = random.Random(6)
    B = 96
    rows = [S.sign(TOY, TN, TK, rnd) for _ in range(B)]
    for i in range(0, B, 7):
        rows[i][2 * TK] ^= 1                                      # invalid signatures among them
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i in (0, 1, 7, 50, 95):
        sig, failed = _eval(cp.flat, rows[i])
        assert failed is None and b.witness(i) == sig and sig[1] == (0 if i % 7 == 0 else 1)
    b.close(); c.close()


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
