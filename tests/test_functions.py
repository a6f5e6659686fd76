"""Tier 2: circom *functions* whose loops and branches depend on run-time values, and arrays indexed by a
run-time value.  The reference emits them as real C++ control flow (loop_bucket.rs:76-91, branch_bucket.rs:100-122,
call_bucket.rs:466-533) with addresses through Fr_toInt (compute_bucket.rs:361-363); here they are a register bytecode
(frontend/rtcode.py) that the oracle interprets, that oracle/emit_ref_cpp.py prints over the reference's own Fr_* calls
(so the reference RUNTIME executes it), and that the HIP kernel interprets per lane (D_CALL, divergent lanes take turns).
"""
import random

import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements.lower import lower
from circom_amd.hip_elements.writers import wtns_bytes
from oracle.tape_eval import eval_flat, eval_tape, check_r1cs


def build_divmod(f, a, b):
    """shift-subtract long division: both loops run a value-dependent number of times (bigint division of circom-ecdsa
    has this shape)"""
    q = f.var(0)
    r = f.var(a)
    sh = f.var(0)
    d = f.var(b)
    with f.loop() as L:                       # align the divisor under the dividend
        L.break_unless((d << 1).leq(r))
        d.set(d << 1)
        sh.set(sh + 1)
    with f.loop() as L:
        with f.if_(r.geq(d)):
            r.set(r - d)
            q.set(q + (f.lift(1) << sh))
        L.break_unless(sh.neq(0))
        d.set(d >> 1)
        sh.set(sh - 1)
    return [q, r]


@template
def LongDiv(c):
    a = c.input("a")
    b = c.input("b")
    qo = c.output("q")
    ro = c.output("r")
    fn = c.function("divmod", 2, build_divmod)
    q, r = c.call(fn, [a, b])
    c.hint(qo, q)
    c.hint(ro, r)
    c.enforce(qo * b + ro, a)


def build_pick(f, *args):
    arr = f.args_array(0, 8)
    sel = args[8]
    acc = f.var(arr.load(sel) * 3 + 1)
    hist = f.array(4)                         # a local array written through a run-time index
    hist.store(sel & 3, acc)
    with f.if_(sel.gt(3)):
        acc.set(acc + hist.load(sel - 4))
    with f.else_():
        acc.set(acc - hist[0])
    return [acc]


@template
def Pick(c):
    arr = c.input("arr", 8)
    sel = c.input("sel")
    out = c.output("out")
    fn = c.function("pick", 9, build_pick)
    (v,) = c.call(fn, [arr[k] for k in range(8)] + [sel])
    c.hint(out, v)


def _pick_model(q, arr, sel):
    acc = (arr[sel] * 3 + 1) % q
    hist = [0, 0, 0, 0]
    hist[sel & 3] = acc
    return (acc + hist[sel - 4]) % q if sel > 3 else (acc - hist[0]) % q


def _flat(fc, inp):
    return eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp, fc.functions)


def test_oracle_interprets_loops_branches_and_indexed_arrays():
    fc = flatten(Program(LongDiv()))
    rnd = random.Random(1)
    for _ in range(300):
        a, b = rnd.randrange(1 << 20), rnd.randrange(1, 1 << 12)
        sig, failed = _flat(fc, {3: a, 4: b})
        assert failed is None and sig[1:3] == [a // b, a % b]
        assert check_r1cs(fc.fp.q, fc.constraints, sig) is None
    fc = flatten(Program(Pick()))
    q = fc.fp.q
    for sel in range(8):
        arr = [rnd.randrange(q) for _ in range(8)]
        inp = {2 + k: arr[k] for k in range(8)}
        inp[10] = sel
        sig, failed = _flat(fc, inp)
        assert failed is None and sig[1] == _pick_model(q, arr, sel)
    inp[10] = 8                                     # outside the array: the reference would read past it
    assert _flat(fc, inp)[1] is not None


@pytest.mark.parametrize("tmpl", [LongDiv, Pick])
def test_lowered_schedule_runs_the_function(tmpl):
    fc = flatten(Program(tmpl()))
    rnd = random.Random(2)
    for S in (1, 4, 16):
        # a call is a heavy unit of its level on ONE strand; the barriers around that level drain global stores (the register
        # window lives in the value table) - eval_tape reports any argument or result that crosses strands without one
        t = lower(fc, n_strands=S)
        assert t.n_strands == S and len(t.functions) == 1
        if S > 1:
            assert t.stats["full_barriers"] >= 2
        for _ in range(20):
            if tmpl is LongDiv:
                inp = {3: rnd.randrange(1 << 20), 4: rnd.randrange(1, 1 << 10)}
            else:
                inp = {2 + k: rnd.randrange(fc.fp.q) for k in range(8)}
                inp[10] = rnd.randrange(8)
            a, fa = _flat(fc, inp)
            b, st = eval_tape(t, inp)
            assert fa is None and st == 0 and a == b


def test_reference_runtime_executes_the_same_function(tmp_path, ref_dir_bn128):
    """the reference's own runtime + field library run the emitted C++ of the function: identical .wtns"""
    from oracle import ref_build
    cp = compile_program(Program(LongDiv()), str(tmp_path), "longdiv", sym=False, strands=(1,))
    ref_build.build_circuit(cp)
    rnd = random.Random(3)
    rows = [[rnd.randrange(1 << 24), rnd.randrange(1, 1 << 12)] for _ in range(20)] + [[5, 7], [0, 3], [(1 << 30) - 1, 1]]
    raw = b"".join(int(v).to_bytes(32, "little") for r in rows for v in r)
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=str(tmp_path / "r_"))
    fc = cp.flat
    for i, (a, b) in enumerate(rows):
        sig, failed = _flat(fc, {3: a, 4: b})
        assert failed is None
        assert (tmp_path / ("r_%d.wtns" % i)).read_bytes() == wtns_bytes(fc.fp.q, sig)
    cp2 = compile_program(Program(Pick()), str(tmp_path), "pickfn", sym=False, strands=(1,))
    ref_build.build_circuit(cp2)
    fc2 = cp2.flat
    rows = [[rnd.randrange(fc2.fp.q) for _ in range(8)] + [sel] for sel in range(8)]
    raw = b"".join(int(v).to_bytes(32, "little") for r in rows for v in r)
    ref_build.run_loop(cp2, raw, len(rows), 1, wtns_prefix=str(tmp_path / "p_"))
    for i, r in enumerate(rows):
        sig, failed = _flat(fc2, {2 + k: v for k, v in enumerate(r)})
        assert failed is None
        assert (tmp_path / ("p_%d.wtns" % i)).read_bytes() == wtns_bytes(fc2.fp.q, sig)


def test_loader_validates_function_bytecode(tmp_path):
    from circom_amd import runtime as rt
    cp = compile_program(Program(Pick()), str(tmp_path), "pickfn", sym=False, strands=(1,))
    rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path).close()
    n_regs, code, _native = cp.tape.functions[0]
    tape = bytearray(open(cp.tape_path, "rb").read())
    blob = code.astype("<u4").tobytes()
    at = bytes(tape).index(blob)
    import numpy as np
    for i, col, val in ((0, 1, n_regs),               # destination register beyond the window
                        (1, 2, n_regs + 5),           # operand register beyond the window
                        (0, 0, 77),                   # unknown opcode
                        (len(code) - 1, 0, 0)):       # no return at the end
        bad = code.copy()
        bad[i, col] = val
        t2 = bytearray(tape)
        t2[at:at + len(blob)] = bad.astype("<u4").tobytes()
        (tmp_path / "bad.cwt").write_bytes(bytes(t2))
        with pytest.raises(rt.CwError):
            rt.Circuit(tmp_path / "bad.cwt", cp.dat_path, cp.r1cs_path)


@pytest.mark.gpu
def test_gpu_interprets_divergent_function_calls(tmp_path):
    from circom_amd import runtime as rt
    cp = compile_program(Program(LongDiv()), str(tmp_path), "longdiv", sym=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    fc = cp.flat
    rnd = random.Random(5)
    B = 200
    rows = [[rnd.randrange(1 << rnd.randrange(1, 40)), rnd.randrange(1, 1 << rnd.randrange(1, 20))] for _ in range(B)]
    rows[7] = [12345, 0]                                    # divisor 0: the alignment loop never ends -> step limit
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    for i, (x, y) in enumerate(rows):
        if y == 0:
            assert st[i] & rt.ST_ARITH, i
            continue
        assert st[i] == 0, (i, st[i])
        assert b.witness(i) == _flat(fc, {3: x, 4: y})[0], i
    b.close(); c.close()


@pytest.mark.gpu
def test_gpu_run_time_indexed_arrays(tmp_path):
    from circom_amd import runtime as rt
    cp = compile_program(Program(Pick()), str(tmp_path), "pickfn", sym=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    fc = cp.flat
    rnd = random.Random(6)
    B = 130
    rows = [[rnd.randrange(c.q) for _ in range(8)] + [rnd.randrange(8)] for _ in range(B)]
    rows[3][8] = 8                                          # index outside the array
    rows[64][8] = c.q - 1                                   # "-1"
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.sync()
    st = b.status()
    for i, r in enumerate(rows):
        if not 0 <= r[8] < 8:
            assert st[i] & rt.ST_ARITH, i
            continue
        assert st[i] == 0
        assert b.witness(i)[1] == _pick_model(c.q, r[:8], r[8]), i
        assert b.witness(i) == _flat(fc, {2 + k: v for k, v in enumerate(r)})[0]
    b.close(); c.close()


# ---- circom-ecdsa-shaped big-integer arithmetic on the BLS12-381 scalar field (BASELINE config 5's building block) ----
def _limbs(x, n, m):
    return [(x >> (n * i)) & ((1 << n) - 1) for i in range(m)]


def _bigmult_rows(n, k, count, seed):
    rnd = random.Random(seed)
    rows = []
    for it in range(count):
        p = (1 << (n * k)) - 1 if it == 0 else rnd.randrange(1 << (n * k - 1), 1 << (n * k))
        a, b = (p - 1, p - 1) if it == 1 else (rnd.randrange(p), rnd.randrange(p))
        rows.append((a, b, p, _limbs(a, n, k) + _limbs(b, n, k) + _limbs(p, n, k)))
    return rows


def test_bigint_mult_mod_p_oracle_and_reference_runtime(tmp_path):
    """a*b mod p on 3 x 32-bit limbs: the witness comes from the circom-ecdsa-style long_div / short_div functions
    (data-dependent branches); the oracle, the lowered schedule and the REFERENCE runtime on bls12381 agree"""
    from circom_amd.circuits.bigint import BigMultModP
    from oracle import ref_build
    import os
    if not os.path.isdir(os.path.join(os.path.dirname(ref_build.__file__), "_ref", "bls12381")) and not ref_build.REF_ROOT.exists():
        pytest.skip("no bls12381 reference build")
    n, k = 32, 3
    cp = compile_program(Program(BigMultModP(n, k), prime="bls12381"), str(tmp_path), "bigmultmodp_bls", sym=False, strands=(1,))
    fc = cp.flat
    rows = _bigmult_rows(n, k, 12, 1)
    for a, b, p, vals in rows:
        inp = {fc.main_input_start + i: v for i, v in enumerate(vals)}
        sig, failed = _flat(fc, inp)
        assert failed is None
        assert sum(sig[1 + i] << (n * i) for i in range(k)) == a * b % p
        assert check_r1cs(fc.fp.q, fc.constraints, sig) is None
    s2, st = eval_tape(cp.tape, {fc.main_input_start + i: v for i, v in enumerate(rows[3][3])})
    assert st == 0 and s2 == _flat(fc, {fc.main_input_start + i: v for i, v in enumerate(rows[3][3])})[0]
    ref_build.build_circuit(cp)
    raw = b"".join(int(v).to_bytes(32, "little") for r in rows for v in r[3])
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=str(tmp_path / "b_"))
    for i, (a, b, p, vals) in enumerate(rows):
        sig, _ = _flat(fc, {fc.main_input_start + j: v for j, v in enumerate(vals)})
        assert (tmp_path / ("b_%d.wtns" % i)).read_bytes() == wtns_bytes(fc.fp.q, sig)


@pytest.mark.gpu
def test_gpu_bigint_mult_mod_p_bls12381(tmp_path):
    from circom_amd import runtime as rt
    from circom_amd.circuits.bigint import BigMultModP
    n, k = 32, 3
    cp = compile_program(Program(BigMultModP(n, k), prime="bls12381"), str(tmp_path), "bigmultmodp_bls", sym=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    fc = cp.flat
    rows = _bigmult_rows(n, k, 150, 2)
    b = c.batch(len(rows))
    b.set_inputs([r[3] for r in rows])
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i, (x, y, p, vals) in enumerate(rows):
        w = b.witness(i)
        assert sum(w[1 + j] << (n * j) for j in range(k)) == x * y % p, i
        if i % 10 == 0:
            assert w == _flat(fc, {fc.main_input_start + j: v for j, v in enumerate(vals)})[0]
    b.close(); c.close()


def test_bytecode_optimiser_keeps_the_function(tmp_path):
    """frontend/rtcode.py folds `t = a op b; v = t` and renumbers registers by live range (794 -> 35 registers, 1266 -> 968
    instructions for long_div(32, 3)): the optimised bytecode computes what the bytecode as written computes, on loops,
    branches, indexed arrays and the long division with its correction branches"""
    from circom_amd.circuits.bigint import BigMultModP
    from oracle.field import Field
    from oracle.tape_eval import run_function
    rnd = random.Random(9)
    for prog, gen in ((Program(LongDiv()), lambda q: [rnd.randrange(1 << 20), rnd.randrange(1, 1 << 12)]),
                      (Program(Pick()), lambda q: [rnd.randrange(q) for _ in range(8)] + [rnd.randrange(8)]),
                      (Program(BigMultModP(32, 3), prime="bls12381"), None)):
        fc = flatten(prog)
        q = fc.fp.q
        f = Field(q)
        for fn_obj, fn in zip(prog.functions, fc.functions):
            assert fn["n_regs"] <= fn_obj.n_regs_built and len(fn["code"]) <= len(fn_obj.code_built)
            built = {"code": [list(c) for c in fn_obj.code_built]}
            cid = {v: i for i, v in enumerate(fc.constants)}
            # the flat form names constants by index; the code as built carries values: intern them the same way
            for c in built["code"]:
                for k in (2, 3):
                    if isinstance(c[k], tuple) and c[k][0] == 'c':
                        c[k] = ('c', cid.setdefault(c[k][1], len(cid)))
            consts = [None] * len(cid)
            for v, i in cid.items():
                consts[i] = v
            for _ in range(60):
                if gen is not None:
                    args = gen(q)[:fn["n_args"]]
                    args += [rnd.randrange(1, 1 << 12) for _ in range(fn["n_args"] - len(args))]
                else:
                    p = rnd.randrange(1 << 95, 1 << 96)
                    a2 = rnd.randrange(p * p)
                    args = _limbs(a2, 32, 6) + _limbs(p, 32, 3)
                r1 = list(args) + [0] * fn_obj.n_regs_built
                r2 = list(args) + [0] * fn["n_regs"]
                ok1 = run_function(f, built, r1, 0, consts)
                ok2 = run_function(f, fn, r2, 0, fc.constants)
                assert ok1 == ok2
                if ok1:
                    assert r1[fn_obj.ret_base_built:fn_obj.ret_base_built + fn["n_ret"]] == r2[fn["ret_base"]:fn["ret_base"] + fn["n_ret"]]
    fn = flatten(Program(BigMultModP(32, 3), prime="bls12381")).functions[0]
    assert fn["n_regs"] < 64 and len(fn["code"]) < 1000


def _random_function_builder(seed, n_args):
    """a random structured function over the builder API of frontend/rtcode.py: mutable vars, nested if / else, bounded
    loops, a local array written and read through run-time indices - every shape the optimiser's liveness has to survive"""
    def build(f, *args):
        rnd = random.Random(seed)
        vars_ = [f.var(a) for a in args] + [f.var(rnd.randrange(1, 50)) for _ in range(3)]
        arr = f.array(4, [rnd.randrange(9) for _ in range(4)])

        def expr(depth=0):
            a, b = rnd.choice(vars_), rnd.choice(vars_)
            k = rnd.randrange(8)
            if k == 0: return a + b
            if k == 1: return a - b
            if k == 2: return a * (rnd.randrange(1, 7))
            if k == 3: return (a & 255) + (b & 15)
            if k == 4: return a.lt(b)
            if k == 5: return arr.load((a & 3))
            if k == 6: return (a >> 1) + 1
            return a.eq(b) + 2

        def block(depth):
            for _ in range(rnd.randrange(2, 5)):
                k = rnd.randrange(7 if depth < 2 else 4)
                if k <= 2:
                    rnd.choice(vars_).set(expr())
                elif k == 3:
                    arr.store(rnd.choice(vars_) & 3, expr())
                elif k == 4:
                    with f.if_((rnd.choice(vars_) & 1).eq(rnd.randrange(2))):
                        block(depth + 1)
                    if rnd.randrange(2):
                        with f.else_():
                            block(depth + 1)
                elif k == 5:
                    cnt = f.var(rnd.randrange(1, 4))
                    with f.loop() as L:
                        L.break_unless(cnt.neq(0))
                        block(depth + 1)
                        cnt.set(cnt - 1)
                else:
                    with f.if_(rnd.choice(vars_).gt(rnd.choice(vars_))):
                        rnd.choice(vars_).set(expr())
        block(0)
        return [vars_[0] + vars_[1], rnd.choice(vars_), arr.load(vars_[2] & 3)]
    return build


def test_bytecode_optimiser_on_random_structured_functions():
    from circom_amd.frontend.rtcode import RtFunction
    from circom_amd.field import Fp, PRIMES
    from oracle.field import Field
    from oracle.tape_eval import run_function
    fp = Fp(PRIMES["bn128"], "bn128")
    f = Field(fp.q)
    shrunk = 0
    for seed in range(60):
        n_args = 2 + seed % 3
        fn = RtFunction("rnd%d" % seed, n_args, _random_function_builder(seed, n_args), fp)
        # constants by value -> a shared table, as Program.register_function does
        consts, cid = [], {}

        def intern(code):
            out = []
            for c in code:
                c = list(c)
                for k in (2, 3):
                    if isinstance(c[k], tuple) and c[k][0] == 'c':
                        c[k] = ('c', cid.setdefault(c[k][1], len(cid)))
                out.append(c)
            return out

        built, opt = {"code": intern(fn.code_built)}, {"code": intern(fn.code)}
        consts = [None] * len(cid)
        for v, i in cid.items():
            consts[i] = v
        assert fn.n_regs <= fn.n_regs_built and len(fn.code) <= len(fn.code_built)
        shrunk += fn.n_regs < fn.n_regs_built
        rnd = random.Random(1000 + seed)
        for _ in range(25):
            args = [rnd.randrange(0, 64) for _ in range(n_args)]
            r1 = args + [0] * fn.n_regs_built
            r2 = args + [0] * fn.n_regs
            ok1 = run_function(f, built, r1, 0, consts)
            ok2 = run_function(f, opt, r2, 0, consts)
            assert ok1 == ok2, seed
            if ok1:
                assert r1[fn.ret_base_built:fn.ret_base_built + fn.n_ret] == r2[fn.ret_base:fn.ret_base + fn.n_ret], seed
    assert shrunk >= 50
